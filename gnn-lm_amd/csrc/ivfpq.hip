// IVF-PQ list scan with LDS-resident ADC look-up tables (see gnnlm_ivfpq_scan_t in include/gnnlm.h): the scoring half
// of the on-device replacement of the reference's faiss CPU search (knn/knn_model.py:87-101; index
// `OPQ64_1024,IVF4096,PQ64`, nprobe 32: gnnlm_scripts/wiki103/find_knn.sh:8-13).
//
// One workgroup per (query, probed list) task.  The query's table (M x 256 floats: 64 KiB at M = 64) is copied to
// LDS once; every thread then scores keys of the list: its code row (M bytes, 16-B loads) and M table look-ups.
// Tasks arrive grouped by list, so the workgroups running at the same time read the same code rows out of L2 / the
// Infinity Cache; a list is read from HBM about once per search batch.  Bound: the LDS look-up rate (M random 4-B
// reads per (query, key) pair).
#include <cstdlib>

#include "kernels.h"

namespace gnnlm {
namespace {

__global__ __launch_bounds__(256) void ivfpq_scan_kernel(gnnlm_ivfpq_scan_t p) {
    extern __shared__ __attribute__((aligned(16))) float lut[];            // [M][256]
    const int tid = threadIdx.x;
    const int64_t task = blockIdx.x;
    const int q = p.task_q[task], slot = p.task_p[task];
    const int64_t list = p.probe_list[(int64_t)q * p.ld_probe + slot];
    const int M = p.M;
    int64_t lo = 0, hi = 0;
    if (list >= 0) { lo = p.list_off[list]; hi = p.list_off[list + 1]; }
    const int64_t len = hi - lo;
    float* oval = nullptr;
    int64_t* oid = nullptr;
    if (!p.tau) {
        oval = p.out_val + (int64_t)q * p.ld_out + (int64_t)(slot - p.p0) * p.seg;
        if (p.out_id) oid = p.out_id + (int64_t)q * p.ld_out + (int64_t)(slot - p.p0) * p.seg;
        for (int64_t j = min(len, (int64_t)p.seg) + tid; j < p.seg; j += 256) {                // beyond the list
            if (p.out_id) oid[j] = -1; else oval[j] = -INFINITY;
        }
    }
    if (len == 0) return;
    {   // the query's table -> LDS (coalesced 16-B pieces); L2 metric: 2 <q'_m, p_mc> - (|p_mc|^2 + 2 <c_l,m, p_mc>) per entry
        const float4* src = reinterpret_cast<const float4*>(p.lut + (int64_t)q * p.ld_lut);
        float4* dst = reinterpret_cast<float4*>(lut);
        if (p.list_term) {
            const float4* lt = reinterpret_cast<const float4*>(p.list_term + list * p.ld_list_term);
            for (int e = tid; e < M * 64; e += 256) {
                const float4 a = src[e], b = lt[e];
                dst[e] = float4{2.f * a.x - b.x, 2.f * a.y - b.y, 2.f * a.z - b.z, 2.f * a.w - b.w};
            }
        } else {
            for (int e = tid; e < M * 64; e += 256) dst[e] = src[e];
        }
    }
    __syncthreads();
    const float bias = p.probe_bias[(int64_t)q * p.ld_probe + slot];
    const float tau = p.tau ? p.tau[q] : 0.f;
    for (int64_t j = tid; j < len; j += 256) {
        const uint4* crow = reinterpret_cast<const uint4*>(p.codes + (lo + j) * M);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;                       // four chains: the look-ups are independent
        for (int c16 = 0; c16 < M / 16; ++c16) {
            const uint4 v = crow[c16];
            const float* t = lut + c16 * 16 * 256;
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s0 += t[(4 * u + 0) * 256 + (w[u] & 255u)];
                s1 += t[(4 * u + 1) * 256 + ((w[u] >> 8) & 255u)];
                s2 += t[(4 * u + 2) * 256 + ((w[u] >> 16) & 255u)];
                s3 += t[(4 * u + 3) * 256 + (w[u] >> 24)];
            }
        }
        const float s = bias + ((s0 + s1) + (s2 + s3));
        if (!p.tau) {
            if (j < p.seg) { oval[j] = s; if (oid) oid[j] = p.ids[lo + j]; }
        } else if (s > tau) {
            const int pos = atomicAdd(&p.cand_cnt[q], 1);
            if (pos < p.cap) {
                p.cand_val[(int64_t)q * p.cap + pos] = s;
                p.cand_id[(int64_t)q * p.cap + pos] = p.ids[lo + j];
            }
        }
    }
}


// Two tasks per workgroup with the two queries' tables interleaved in LDS (float2 per entry, 128 KiB at M = 64):
// when both tasks probe the SAME list -- the common case, tasks arrive sorted by list -- every key's code row is read
// once and every look-up is one ds_read_b64 that serves both queries: half the LDS instructions and half the address
// arithmetic per (query, key) pair of the one-task kernel above (measured at the reference's index shape, 8192 queries,
// k = 1024, nprobe 32 over 103 M keys: 121 ms of scan time with one query per workgroup).  Tasks of different lists
// (a list boundary inside the pair) are scanned one after the other with their half of the table.
#ifndef GNNLM_IVF_EXP
#define GNNLM_IVF_EXP 0      // ablation builds (tools/build_variant.sh): 1 no code loads, 2 no table load, 4 no outputs
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WHICH>    // 0: query 0 only, 1: query 1 only, 2: both (same list)
__device__ __forceinline__ void scan_list(const gnnlm_ivfpq_scan_t& p, const f32x2* lut, int M, int64_t lo, int64_t len, int tid, int nt,
                                          int q0, int slot0, int q1, int slot1) {
    const float bias0 = WHICH != 1 ? p.probe_bias[(int64_t)q0 * p.ld_probe + slot0] : 0.f;
    const float bias1 = WHICH != 0 ? p.probe_bias[(int64_t)q1 * p.ld_probe + slot1] : 0.f;
    float *ov0 = nullptr, *ov1 = nullptr;
    int64_t *oi0 = nullptr, *oi1 = nullptr;
    float tau0 = 0.f, tau1 = 0.f;
    if (!p.tau) {
        if (WHICH != 1) { ov0 = p.out_val + (int64_t)q0 * p.ld_out + (int64_t)(slot0 - p.p0) * p.seg; if (p.out_id) oi0 = p.out_id + (int64_t)q0 * p.ld_out + (int64_t)(slot0 - p.p0) * p.seg; }
        if (WHICH != 0) { ov1 = p.out_val + (int64_t)q1 * p.ld_out + (int64_t)(slot1 - p.p0) * p.seg; if (p.out_id) oi1 = p.out_id + (int64_t)q1 * p.ld_out + (int64_t)(slot1 - p.p0) * p.seg; }
        for (int64_t j = min(len, (int64_t)p.seg) + tid; j < p.seg; j += nt) {
            if (WHICH != 1) { if (oi0) oi0[j] = -1; else ov0[j] = -INFINITY; }
            if (WHICH != 0) { if (oi1) oi1[j] = -1; else ov1[j] = -INFINITY; }
        }
    } else {
        if (WHICH != 1) tau0 = p.tau[q0];
        if (WHICH != 0) tau1 = p.tau[q1];
    }
    for (int64_t j = tid; j < len; j += nt) {
        const uint4* crow = reinterpret_cast<const uint4*>(p.codes + (lo + j) * M);
        f32x2 s0 = {0.f, 0.f}, s1 = {0.f, 0.f}, s2 = {0.f, 0.f}, s3 = {0.f, 0.f};
        for (int c16 = 0; c16 < M / 16; ++c16) {
            const uint4 v = crow[c16];
            const f32x2* t = lut + c16 * 16 * 256;
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s0 += t[(4 * u + 0) * 256 + (w[u] & 255u)];
                s1 += t[(4 * u + 1) * 256 + ((w[u] >> 8) & 255u)];
                s2 += t[(4 * u + 2) * 256 + ((w[u] >> 16) & 255u)];
                s3 += t[(4 * u + 3) * 256 + (w[u] >> 24)];
            }
        }
        const f32x2 s = (s0 + s1) + (s2 + s3);
        if (!p.tau) {
            if (j < p.seg) {
                if (WHICH != 1) ov0[j] = bias0 + s.x;
                if (WHICH != 0) ov1[j] = bias1 + s.y;
                if (p.out_id) {
                    const int64_t id = p.ids[lo + j];
                    if (WHICH != 1) oi0[j] = id;
                    if (WHICH != 0) oi1[j] = id;
                }
            }
        } else {
            const int64_t id = p.ids[lo + j];
            if (WHICH != 1 && bias0 + s.x > tau0) {
                const int pos = atomicAdd(&p.cand_cnt[q0], 1);
                if (pos < p.cap) { p.cand_val[(int64_t)q0 * p.cap + pos] = bias0 + s.x; p.cand_id[(int64_t)q0 * p.cap + pos] = id; }
            }
            if (WHICH != 0 && bias1 + s.y > tau1) {
                const int pos = atomicAdd(&p.cand_cnt[q1], 1);
                if (pos < p.cap) { p.cand_val[(int64_t)q1 * p.cap + pos] = bias1 + s.y; p.cand_id[(int64_t)q1 * p.cap + pos] = id; }
            }
        }
    }
}

__global__ __launch_bounds__(1024) void ivfpq_scan2_kernel(gnnlm_ivfpq_scan_t p) {
    extern __shared__ __attribute__((aligned(16))) float lut_raw[];
    f32x2* lut = reinterpret_cast<f32x2*>(lut_raw);                       // [M][256] (query 0, query 1)
    const int tid = threadIdx.x, nt = 1024;
    const int64_t t0 = 2 * (int64_t)blockIdx.x, t1 = t0 + 1;
    const bool two = t1 < p.n_tasks;
    const int M = p.M;
    const int q0 = p.task_q[t0], slot0 = p.task_p[t0];
    const int q1 = two ? p.task_q[t1] : q0, slot1 = two ? p.task_p[t1] : slot0;
    const int64_t la = p.probe_list[(int64_t)q0 * p.ld_probe + slot0];
    const int64_t lb = two ? p.probe_list[(int64_t)q1 * p.ld_probe + slot1] : -1;
    {
        const float* a = p.lut + (int64_t)q0 * p.ld_lut;
        const float* b = p.lut + (int64_t)q1 * p.ld_lut;
        for (int e = tid; e < M * 256; e += nt) lut[e] = f32x2{a[e], b[e]};
    }
    __syncthreads();
    int64_t loa = 0, lena = 0, lob = 0, lenb = 0;
    if (la >= 0) { loa = p.list_off[la]; lena = p.list_off[la + 1] - loa; }
    if (lb >= 0) { lob = p.list_off[lb]; lenb = p.list_off[lb + 1] - lob; }
    if (two && la == lb) {
        scan_list<2>(p, lut, M, loa, lena, tid, nt, q0, slot0, q1, slot1);
    } else {
        scan_list<0>(p, lut, M, loa, lena, tid, nt, q0, slot0, q1, slot1);
        if (two) scan_list<1>(p, lut, M, lob, lenb, tid, nt, q0, slot0, q1, slot1);
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Conflict-free scan over the PACKED index image (gnnlm_ivfpq_pack_codes / gnnlm_ivfpq_pack_lut), M = 32 or 64.
//
// The kernels above are bound by LDS bank conflicts: the 32 lanes of a half-wave read 32 random entries of ONE
// sub-quantizer's table (~3.5 lanes on the busiest bank pair).  Here a lane's look-up number s goes to sub-quantizer
// (lane + s) mod 32 of its own key, and the tables are stored [half][code][32 sub-quantizers] (8 B per entry: the two
// queries of the workgroup), so the 32 lanes of a group always hit 32 different bank pairs whatever the codes are.  The
// key bytes are stored in that rotated order (byte s of a half = sub-quantizer (row + s) mod 32), so the lane still
// extracts fixed byte positions, and the LDS address of a look-up is ONE v_perm_b32: {0, half, code byte, slot offset}.
// Per (two queries, key, sub-quantizer): v_perm_b32 + ds_read_b64 + v_pk_add_f32.  Key rows are stored piece-major in
// blocks of 64 rows, so a wave's 16-B loads are 1 KiB contiguous.  Workgroups of one XCD take neighbouring task pairs
// (one list's tasks share an L2).
constexpr int STAGE_CAP = 1024;
constexpr int STAGE_BYTES = 16 + 2 * STAGE_CAP * 8;
template <int M, int WHICH>    // WHICH as above
__device__ __forceinline__ void scan_rot(const gnnlm_ivfpq_scan_t& p, const char* lut, int* stage, int64_t lo, int64_t len, int tid,
                                         int q0, int slot0, int q1, int slot1) {
    constexpr int NT = 1024, NW = NT / 64;
    // filtered mode: survivors are collected in LDS ({score, row} pairs, STAGE_CAP per query) and written out once per
    // task with ONE global atomic per query -- no device-scope atomic (and its vmcnt(0)) inside the scan loop
    int* scnt = stage;                                            // [2] survivors per query, then [2] output bases
    float2* sbuf = reinterpret_cast<float2*>(stage + 4);          // [2][STAGE_CAP] {score, row - lo as int bits}
    if (p.tau) {
        if (tid < 2) scnt[tid] = 0;
        __syncthreads();
    }
    const int lane = tid & 63, wave = tid >> 6;
    const float bias0 = WHICH != 1 ? p.probe_bias[(int64_t)q0 * p.ld_probe + slot0] : 0.f;
    const float bias1 = WHICH != 0 ? p.probe_bias[(int64_t)q1 * p.ld_probe + slot1] : 0.f;
    float *ov0 = nullptr, *ov1 = nullptr;
    int64_t *oi0 = nullptr, *oi1 = nullptr;
    float tau0 = 0.f, tau1 = 0.f;
    if (!p.tau) {
        if (WHICH != 1) { ov0 = p.out_val + (int64_t)q0 * p.ld_out + (int64_t)(slot0 - p.p0) * p.seg; if (p.out_id) oi0 = p.out_id + (int64_t)q0 * p.ld_out + (int64_t)(slot0 - p.p0) * p.seg; }
        if (WHICH != 0) { ov1 = p.out_val + (int64_t)q1 * p.ld_out + (int64_t)(slot1 - p.p0) * p.seg; if (p.out_id) oi1 = p.out_id + (int64_t)q1 * p.ld_out + (int64_t)(slot1 - p.p0) * p.seg; }
        for (int64_t j = min(len, (int64_t)p.seg) + tid; j < p.seg; j += NT) {
            if (WHICH != 1) { if (oi0) oi0[j] = -1; else ov0[j] = -INFINITY; }
            if (WHICH != 0) { if (oi1) oi1[j] = -1; else ov1[j] = -INFINITY; }
        }
    } else {
        if (WHICH != 1) tau0 = p.tau[q0];
        if (WHICH != 0) tau1 = p.tau[q1];
    }
    // slot offsets of two look-ups per register (8 * ((lane + s) mod 32)) and a 0x01 byte that becomes bit 16 of the
    // address for the second half's tables (+64 KiB)
    uint32_t t[16];
#pragma unroll
    for (int g = 0; g < 16; ++g)
        t[g] = (uint32_t)(((lane + 2 * g) & 31) << 3) | (uint32_t)(((lane + 2 * g + 1) & 31) << 3) << 8 | 0x00010000u;
    const int64_t hi = lo + len;
    const int64_t b_end = (hi + 63) >> 6;
    constexpr int NH = M / 32;
    uint32_t w[NH][8];                          // the lane's row, one half (two 16-B pieces) per register set
    const int64_t b_first = (lo >> 6) + wave;
    auto load_half = [&](int64_t blk, int h) {
        const uint4* src = reinterpret_cast<const uint4*>(p.codes + blk * (64 * M)) + lane;
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
#if GNNLM_IVF_EXP & 1
            const uint32_t z = (uint32_t)blk * 2654435761u + lane * 40503u + pc + 2 * h;
            const uint4 v = uint4{z, z * 3u, z * 5u, z * 7u};
            (void)src;
#else
            const uint4 v = src[(2 * h + pc) * 64];
#endif
            w[h][4 * pc] = v.x; w[h][4 * pc + 1] = v.y; w[h][4 * pc + 2] = v.z; w[h][4 * pc + 3] = v.w;
        }
    };
    if (b_first < b_end) {
#pragma unroll
        for (int h = 0; h < NH; ++h) load_half(b_first, h);
    }
    for (int64_t b = b_first; b < b_end; b += NW) {
        const int64_t bn = min(b + NW, b_end - 1);      // the next block of this wave (last: re-read)
        f32x2 acc[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            // 32 look-ups, eight reads in flight: v_perm_b32 builds the address in the low register of the pair the read
            // returns to ({0, h (the 0x01 of t), code byte, slot offset}); the sums run eight look-ups behind
            const uint32_t sh = h ? 0x0c020000u : 0x0c0c0000u;
            const uint32_t s0 = sh | 0x0400u, s1 = sh | 0x0501u, s2 = sh | 0x0600u, s3 = sh | 0x0701u;
            asm volatile(
                "v_perm_b32 v112, %2, %10, %26\n"
                "ds_read_b64 v[112:113], v112\n"
                "v_perm_b32 v114, %2, %10, %27\n"
                "ds_read_b64 v[114:115], v114\n"
                "v_perm_b32 v116, %2, %11, %28\n"
                "ds_read_b64 v[116:117], v116\n"
                "v_perm_b32 v118, %2, %11, %29\n"
                "ds_read_b64 v[118:119], v118\n"
                "v_perm_b32 v120, %3, %12, %26\n"
                "ds_read_b64 v[120:121], v120\n"
                "v_perm_b32 v122, %3, %12, %27\n"
                "ds_read_b64 v[122:123], v122\n"
                "v_perm_b32 v124, %3, %13, %28\n"
                "ds_read_b64 v[124:125], v124\n"
                "v_perm_b32 v126, %3, %13, %29\n"
                "ds_read_b64 v[126:127], v126\n"
                "s_waitcnt lgkmcnt(7)\n"
                "v_pk_add_f32 %0, %0, v[112:113]\n"
                "v_perm_b32 v112, %4, %14, %26\n"
                "ds_read_b64 v[112:113], v112\n"
                "s_waitcnt lgkmcnt(7)\n"
                "v_pk_add_f32 %1, %1, v[114:115]\n"
                "v_perm_b32 v114, %4, %14, %27\n"
                "ds_read_b64 v[114:115], v114\n"
                "s_waitcnt lgkmcnt(7)\n"
                "v_pk_add_f32 %0, %0, v[116:117]\n"
                "v_perm_b32 v116, %4, %15, %28\n"
                "ds_read_b64 v[116:117], v116\n"
                "s_waitcnt lgkmcnt(7)\n"
                "v_pk_add_f32 %1, %1, v[118:119]\n"
                "v_perm_b32 v118, %4, %15, %29\n"
                "ds_read_b64 v[118:119], v118\n"
                "s_waitcnt lgkmcnt(7)\n"
                "v_pk_add_f32 %0, %0, v[120:121]\n"
                "v_perm_b32 v120, %5, %16, %26\n"
                "ds_read_b64 v[120:121], v120\n"
                "s_waitcnt lgkmcnt(7)\n"
                "v_pk_add_f32 %1, %1, v[122:123]\n"
                "v_perm_b32 v122, %5, %16, %27\n"
                "ds_read_b64 v[122:123], v122\n"
                "s_waitcnt lgkmcnt(7)\n"
                "v_pk_add_f32 %0, %0, v[124:125]\n"
                "v_perm_b32 v124, %5, %17, %28\n"
                "ds_read_b64 v[124:125], v124\n"
                "s_waitcnt lgkmcnt(7)\n"
                "v_pk_add_f32 %1, %1, v[126:127]\n"
                "v_perm_b32 v126, %5, %17, %29\n"
                "ds_read_b64 v[126:127], v126\n"
                "s_waitcnt lgkmcnt(7)\n"
                "v_pk_add_f32 %0, %0, v[112:113]\n"
                "v_perm_b32 v112, %6, %18, %26\n"
                "ds_read_b64 v[112:113], v112\n"
                "s_waitcnt lgkmcnt(7)\n"
                "v_pk_add_f32 %1, %1, v[114:115]\n"
                "v_perm_b32 v114, %6, %18, %27\n"
                "ds_read_b64 v[114:115], v114\n"
                "s_waitcnt lgkmcnt(7)\n"
                "v_pk_add_f32 %0, %0, v[116:117]\n"
                "v_perm_b32 v116, %6, %19, %28\n"
                "ds_read_b64 v[116:117], v116\n"
                "s_waitcnt lgkmcnt(7)\n"
                "v_pk_add_f32 %1, %1, v[118:119]\n"
                "v_perm_b32 v118, %6, %19, %29\n"
                "ds_read_b64 v[118:119], v118\n"
                "s_waitcnt lgkmcnt(7)\n"
                "v_pk_add_f32 %0, %0, v[120:121]\n"
                "v_perm_b32 v120, %7, %20, %26\n"
                "ds_read_b64 v[120:121], v120\n"
                "s_waitcnt lgkmcnt(7)\n"
                "v_pk_add_f32 %1, %1, v[122:123]\n"
                "v_perm_b32 v122, %7, %20, %27\n"
                "ds_read_b64 v[122:123], v122\n"
                "s_waitcnt lgkmcnt(7)\n"
                "v_pk_add_f32 %0, %0, v[124:125]\n"
                "v_perm_b32 v124, %7, %21, %28\n"
                "ds_read_b64 v[124:125], v124\n"
                "s_waitcnt lgkmcnt(7)\n"
                "v_pk_add_f32 %1, %1, v[126:127]\n"
                "v_perm_b32 v126, %7, %21, %29\n"
                "ds_read_b64 v[126:127], v126\n"
                "s_waitcnt lgkmcnt(7)\n"
                "v_pk_add_f32 %0, %0, v[112:113]\n"
                "v_perm_b32 v112, %8, %22, %26\n"
                "ds_read_b64 v[112:113], v112\n"
                "s_waitcnt lgkmcnt(7)\n"
                "v_pk_add_f32 %1, %1, v[114:115]\n"
                "v_perm_b32 v114, %8, %22, %27\n"
                "ds_read_b64 v[114:115], v114\n"
                "s_waitcnt lgkmcnt(7)\n"
                "v_pk_add_f32 %0, %0, v[116:117]\n"
                "v_perm_b32 v116, %8, %23, %28\n"
                "ds_read_b64 v[116:117], v116\n"
                "s_waitcnt lgkmcnt(7)\n"
                "v_pk_add_f32 %1, %1, v[118:119]\n"
                "v_perm_b32 v118, %8, %23, %29\n"
                "ds_read_b64 v[118:119], v118\n"
                "s_waitcnt lgkmcnt(7)\n"
                "v_pk_add_f32 %0, %0, v[120:121]\n"
                "v_perm_b32 v120, %9, %24, %26\n"
                "ds_read_b64 v[120:121], v120\n"
                "s_waitcnt lgkmcnt(7)\n"
                "v_pk_add_f32 %1, %1, v[122:123]\n"
                "v_perm_b32 v122, %9, %24, %27\n"
                "ds_read_b64 v[122:123], v122\n"
                "s_waitcnt lgkmcnt(7)\n"
                "v_pk_add_f32 %0, %0, v[124:125]\n"
                "v_perm_b32 v124, %9, %25, %28\n"
                "ds_read_b64 v[124:125], v124\n"
                "s_waitcnt lgkmcnt(7)\n"
                "v_pk_add_f32 %1, %1, v[126:127]\n"
                "v_perm_b32 v126, %9, %25, %29\n"
                "ds_read_b64 v[126:127], v126\n"
                "s_waitcnt lgkmcnt(7)\n"
                "v_pk_add_f32 %0, %0, v[112:113]\n"
                "s_waitcnt lgkmcnt(6)\n"
                "v_pk_add_f32 %1, %1, v[114:115]\n"
                "s_waitcnt lgkmcnt(5)\n"
                "v_pk_add_f32 %0, %0, v[116:117]\n"
                "s_waitcnt lgkmcnt(4)\n"
                "v_pk_add_f32 %1, %1, v[118:119]\n"
                "s_waitcnt lgkmcnt(3)\n"
                "v_pk_add_f32 %0, %0, v[120:121]\n"
                "s_waitcnt lgkmcnt(2)\n"
                "v_pk_add_f32 %1, %1, v[122:123]\n"
                "s_waitcnt lgkmcnt(1)\n"
                "v_pk_add_f32 %0, %0, v[124:125]\n"
                "s_waitcnt lgkmcnt(0)\n"
                "v_pk_add_f32 %1, %1, v[126:127]\n"
                : "+v"(acc[0]), "+v"(acc[1])
                : "v"(w[h][0]), "v"(w[h][1]), "v"(w[h][2]), "v"(w[h][3]), "v"(w[h][4]), "v"(w[h][5]), "v"(w[h][6]), "v"(w[h][7]),
                  "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]), "v"(t[5]), "v"(t[6]), "v"(t[7]),
                  "v"(t[8]), "v"(t[9]), "v"(t[10]), "v"(t[11]), "v"(t[12]), "v"(t[13]), "v"(t[14]), "v"(t[15]),
                  "s"(s0), "s"(s1), "s"(s2), "s"(s3)
                : "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125",
                  "v126", "v127");
            load_half(bn, h);                  // the next block's half travels under the other half's look-ups
        }
        const f32x2 sum = acc[0] + acc[1];
        const int64_t r = b * 64 + lane;
#if GNNLM_IVF_EXP & 4
        if (sum.x != 1234.5f) continue;
#endif
        if (r < lo || r >= hi) continue;
        const int64_t j = r - lo;
        if (!p.tau) {
            if (j < p.seg) {
                if (WHICH != 1) ov0[j] = bias0 + sum.x;
                if (WHICH != 0) ov1[j] = bias1 + sum.y;
                if (p.out_id) {
                    const int64_t id = p.ids[r];
                    if (WHICH != 1) oi0[j] = id;
                    if (WHICH != 0) oi1[j] = id;
                }
            }
        } else {
            if (WHICH != 1 && bias0 + sum.x > tau0) {
                const int pos = atomicAdd(&scnt[0], 1);
                if (pos < STAGE_CAP) sbuf[pos] = float2{bias0 + sum.x, __int_as_float((int)j)};
                else {                                              // staging full: straight to the candidate rows
                    const int gp = atomicAdd(&p.cand_cnt[q0], 1);
                    if (gp < p.cap) { p.cand_val[(int64_t)q0 * p.cap + gp] = bias0 + sum.x; p.cand_id[(int64_t)q0 * p.cap + gp] = p.ids[r]; }
                }
            }
            if (WHICH != 0 && bias1 + sum.y > tau1) {
                const int pos = atomicAdd(&scnt[1], 1);
                if (pos < STAGE_CAP) sbuf[STAGE_CAP + pos] = float2{bias1 + sum.y, __int_as_float((int)j)};
                else {
                    const int gp = atomicAdd(&p.cand_cnt[q1], 1);
                    if (gp < p.cap) { p.cand_val[(int64_t)q1 * p.cap + gp] = bias1 + sum.y; p.cand_id[(int64_t)q1 * p.cap + gp] = p.ids[r]; }
                }
            }
        }
    }
    if (p.tau) {
        __syncthreads();
        if (tid < 2 && (WHICH == 2 || WHICH == tid)) {
            const int n = min(scnt[tid], STAGE_CAP);
            scnt[2 + tid] = n ? atomicAdd(&p.cand_cnt[tid ? q1 : q0], n) : 0;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (WHICH != 2 && WHICH != u) continue;
            const int n = min(scnt[u], STAGE_CAP), base = scnt[2 + u];
            const int q = u ? q1 : q0;
            for (int e = tid; e < n; e += NT) {
                const float2 c = sbuf[u * STAGE_CAP + e];
                if (base + e < p.cap) {
                    p.cand_val[(int64_t)q * p.cap + base + e] = c.x;
                    p.cand_id[(int64_t)q * p.cap + base + e] = p.ids[lo + __float_as_int(c.y)];
                }
            }
        }
        __syncthreads();                       // the staging area is reused by the next scan of this workgroup
    }
}

template <int M>
__global__ __launch_bounds__(1024) void ivfpq_scan_rot_kernel(gnnlm_ivfpq_scan_t p, int64_t n_pairs, int pairs_per_xcd) {
    extern __shared__ __attribute__((aligned(16))) float lut_raw[];
    f32x2* lut = reinterpret_cast<f32x2*>(lut_raw);                       // [M/32][256][32] (query 0, query 1)
    const int tid = threadIdx.x;
    // consecutive workgroups go to consecutive XCDs: give each XCD a contiguous range of the list-sorted task pairs
    const int64_t pair = (int64_t)(blockIdx.x & 7) * pairs_per_xcd + (blockIdx.x >> 3);
    if (pair >= n_pairs) return;
    const int64_t t0 = 2 * pair, t1 = t0 + 1;
    const bool two = t1 < p.n_tasks;
    const int q0 = p.task_q[t0], slot0 = p.task_p[t0];
    const int q1 = two ? p.task_q[t1] : q0, slot1 = two ? p.task_p[t1] : slot0;
    const int64_t la = p.probe_list[(int64_t)q0 * p.ld_probe + slot0];
    const int64_t lb = two ? p.probe_list[(int64_t)q1 * p.ld_probe + slot1] : -1;
    {
        const float4* a = reinterpret_cast<const float4*>(p.lut + (int64_t)q0 * p.ld_lut);
        const float4* b = reinterpret_cast<const float4*>(p.lut + (int64_t)q1 * p.ld_lut);
        f32x4* dst = reinterpret_cast<f32x4*>(lut_raw);
        for (int e = tid; e < M * 64; e += 1024) {
#if GNNLM_IVF_EXP & 2
            if (a[0].x != 1234.5f) break;
#endif
            const float4 x = a[e], y = b[e];
            dst[2 * e] = f32x4{x.x, y.x, x.y, y.y};
            dst[2 * e + 1] = f32x4{x.z, y.z, x.w, y.w};
        }
    }
    __syncthreads();
    int64_t loa = 0, lena = 0, lob = 0, lenb = 0;
    if (la >= 0) { loa = p.list_off[la]; lena = p.list_off[la + 1] - loa; }
    if (lb >= 0) { lob = p.list_off[lb]; lenb = p.list_off[lb + 1] - lob; }
    const char* lc = reinterpret_cast<const char*>(lut);
    int* stage = reinterpret_cast<int*>(lut_raw + M * 256 * 2);          // behind the tables: 16 B + 2 x STAGE_CAP x 8 B
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lut_raw != 0u) __builtin_trap();   // look-up addresses are absolute
    if (two && la == lb) {
        scan_rot<M, 2>(p, lc, stage, loa, lena, tid, q0, slot0, q1, slot1);
    } else {
        scan_rot<M, 0>(p, lc, stage, loa, lena, tid, q0, slot0, q1, slot1);
        if (two) scan_rot<M, 1>(p, lc, stage, lob, lenb, tid, q0, slot0, q1, slot1);
    }
}

// codes [N, M] row-major -> packed image: blocks of 64 rows, [M/16 pieces][64 rows][16 B], byte s of half h of row r =
// code[r][32 h + (r + s) mod 32]; rows beyond N are zero.  One thread per (row, piece).
__global__ __launch_bounds__(256) void ivfpq_pack_codes_kernel(const uint8_t* __restrict__ codes, int64_t N, int M, uint8_t* __restrict__ out) {
    const int pieces = M / 16;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = g / pieces;
    const int pc = (int)(g % pieces);
    if (r >= ((N + 63) >> 6 << 6)) return;
    uint32_t v[4] = {0u, 0u, 0u, 0u};
    if (r < N) {
        const uint8_t* row = codes + r * M;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int idx = pc * 16 + i, h = idx >> 5, s = idx & 31;
            v[i >> 2] |= (uint32_t)row[32 * h + (int)((r + s) & 31)] << (8 * (i & 3));
        }
    }
    uint4* dst = reinterpret_cast<uint4*>(out + (r >> 6) * (64 * (int64_t)M) + pc * 1024 + (r & 63) * 16);
    *dst = uint4{v[0], v[1], v[2], v[3]};
}

// lut [n, M, 256] -> [n, M/32, 256, 32]; one workgroup per (query, half)
__global__ __launch_bounds__(256) void ivfpq_pack_lut_kernel(const float* __restrict__ lut, int64_t ld, int M, float* __restrict__ out) {
    __shared__ float tile[32][257];
    const int q = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
    const float* src = lut + (int64_t)q * ld + (int64_t)h * 32 * 256;
    for (int e = tid; e < 32 * 256; e += 256) tile[e >> 8][e & 255] = src[e];
    __syncthreads();
    float* dst = out + ((int64_t)q * (M / 32) + h) * (256 * 32);
    for (int e = tid; e < 32 * 256; e += 256) dst[e] = tile[e & 31][e >> 5];
}

}  // namespace

int ivfpq_pack_codes(const uint8_t* codes, int64_t N, int M, uint8_t* out, hipStream_t stream) {
    GNNLM_REQUIRE(codes && out && N >= 0 && (M == 32 || M == 64), "ivfpq_pack_codes: need M = 32 or 64");
    GNNLM_REQUIRE((uintptr_t)out % 16 == 0, "ivfpq_pack_codes: 16-byte aligned output");
    const int64_t threads = ((N + 63) >> 6 << 6) * (M / 16);
    if (threads == 0) return OK;
    GNNLM_REQUIRE(cdiv(threads, (int64_t)256) < (1ll << 31), "ivfpq_pack_codes: too many rows for one launch");
    hipLaunchKernelGGL(ivfpq_pack_codes_kernel, dim3((unsigned)cdiv(threads, (int64_t)256)), dim3(256), 0, stream, codes, N, M, out);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int ivfpq_pack_lut(const float* lut, int64_t ld_lut, int64_t n, int M, float* out, hipStream_t stream) {
    GNNLM_REQUIRE(lut && out && n >= 0 && n < (1ll << 31) && (M == 32 || M == 64) && ld_lut >= (int64_t)M * 256, "ivfpq_pack_lut: need M = 32 or 64");
    if (n == 0) return OK;
    hipLaunchKernelGGL(ivfpq_pack_lut_kernel, dim3((unsigned)n, (unsigned)(M / 32)), dim3(256), 0, stream, lut, ld_lut, M, out);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int ivfpq_scan(const gnnlm_ivfpq_scan_t& d, hipStream_t stream) {
    GNNLM_REQUIRE(d.n_tasks >= 0 && d.n_tasks < (1ll << 31), "ivfpq_scan: bad task count");
    if (d.n_tasks == 0) return OK;
    GNNLM_REQUIRE(d.codes && d.ids && d.list_off && d.lut && d.probe_list && d.probe_bias && d.task_q && d.task_p,
                  "ivfpq_scan: null operand");
    GNNLM_REQUIRE(d.M > 0 && d.M % 16 == 0 && d.M <= 128 && d.ld_lut >= (int64_t)d.M * 256 && d.ld_lut % 4 == 0 &&
                      (uintptr_t)d.lut % 16 == 0 && (uintptr_t)d.codes % 16 == 0,
                  "ivfpq_scan: need M % 16 == 0, M <= 128, 16-byte aligned tables");
    if (d.tau) GNNLM_REQUIRE(d.cand_val && d.cand_id && d.cand_cnt && d.cap > 0, "ivfpq_scan: filtered mode needs the candidate buffers");
    else GNNLM_REQUIRE(d.out_val && d.seg > 0 && d.ld_out >= d.seg, "ivfpq_scan: dense mode needs the output rows");
    const size_t lds = (size_t)d.M * 256 * sizeof(float);
    GNNLM_LDS_OPT_IN(&ivfpq_scan_kernel, 128 * 1024);
    GNNLM_REQUIRE(!d.list_term || (!d.packed && d.ld_list_term >= (int64_t)d.M * 256 && d.ld_list_term % 4 == 0 && (uintptr_t)d.list_term % 16 == 0),
                  "ivfpq_scan: the L2 metric runs on row-major codes with 16-byte aligned list tables");
    ProfScope prof(K_IVF, stream, 0.0, 0.0);
    if (d.list_term) {
        hipLaunchKernelGGL(ivfpq_scan_kernel, dim3((unsigned)d.n_tasks), dim3(256), lds, stream, d);
        GNNLM_LAUNCH_CHECK();
        return OK;
    }
    if (d.packed) {
        GNNLM_REQUIRE(d.M == 32 || d.M == 64, "ivfpq_scan: the packed image exists for M = 32 and 64");
        const int64_t n_pairs = cdiv(d.n_tasks, (int64_t)2);
        const int per_xcd = (int)cdiv(n_pairs, (int64_t)8);
        GNNLM_LDS_OPT_IN(&ivfpq_scan_rot_kernel<64>, 128 * 1024 + STAGE_BYTES);
        GNNLM_LDS_OPT_IN(&ivfpq_scan_rot_kernel<32>, 64 * 1024 + STAGE_BYTES);
        if (d.M == 64) hipLaunchKernelGGL(ivfpq_scan_rot_kernel<64>, dim3((unsigned)(8 * per_xcd)), dim3(1024), 2 * lds + STAGE_BYTES, stream, d, n_pairs, per_xcd);
        else hipLaunchKernelGGL(ivfpq_scan_rot_kernel<32>, dim3((unsigned)(8 * per_xcd)), dim3(1024), 2 * lds + STAGE_BYTES, stream, d, n_pairs, per_xcd);
        GNNLM_LAUNCH_CHECK();
        return OK;
    }
    if (d.M <= 64 && !getenv("GNNLM_IVF_SINGLE")) {
        GNNLM_LDS_OPT_IN(&ivfpq_scan2_kernel, 128 * 1024);
        hipLaunchKernelGGL(ivfpq_scan2_kernel, dim3((unsigned)cdiv(d.n_tasks, 2)), dim3(1024), 2 * lds, stream, d);
        GNNLM_LAUNCH_CHECK();
        return OK;
    }
    hipLaunchKernelGGL(ivfpq_scan_kernel, dim3((unsigned)d.n_tasks), dim3(256), lds, stream, d);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

}  // namespace gnnlm
