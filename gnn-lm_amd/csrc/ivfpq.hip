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
        oid = p.out_id + (int64_t)q * p.ld_out + (int64_t)(slot - p.p0) * p.seg;
        for (int64_t j = min(len, (int64_t)p.seg) + tid; j < p.seg; j += 256) oid[j] = -1;     // beyond the list
    }
    if (len == 0) return;
    {   // the query's table -> LDS (coalesced 16-B pieces)
        const float4* src = reinterpret_cast<const float4*>(p.lut + (int64_t)q * p.ld_lut);
        float4* dst = reinterpret_cast<float4*>(lut);
        for (int e = tid; e < M * 64; e += 256) dst[e] = src[e];
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
            if (j < p.seg) { oval[j] = s; oid[j] = p.ids[lo + j]; }
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
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int WHICH>    // 0: query 0 only, 1: query 1 only, 2: both (same list)
__device__ __forceinline__ void scan_list(const gnnlm_ivfpq_scan_t& p, const f32x2* lut, int M, int64_t lo, int64_t len, int tid, int nt,
                                          int q0, int slot0, int q1, int slot1) {
    const float bias0 = WHICH != 1 ? p.probe_bias[(int64_t)q0 * p.ld_probe + slot0] : 0.f;
    const float bias1 = WHICH != 0 ? p.probe_bias[(int64_t)q1 * p.ld_probe + slot1] : 0.f;
    float *ov0 = nullptr, *ov1 = nullptr;
    int64_t *oi0 = nullptr, *oi1 = nullptr;
    float tau0 = 0.f, tau1 = 0.f;
    if (!p.tau) {
        if (WHICH != 1) { ov0 = p.out_val + (int64_t)q0 * p.ld_out + (int64_t)(slot0 - p.p0) * p.seg; oi0 = p.out_id + (int64_t)q0 * p.ld_out + (int64_t)(slot0 - p.p0) * p.seg; }
        if (WHICH != 0) { ov1 = p.out_val + (int64_t)q1 * p.ld_out + (int64_t)(slot1 - p.p0) * p.seg; oi1 = p.out_id + (int64_t)q1 * p.ld_out + (int64_t)(slot1 - p.p0) * p.seg; }
        for (int64_t j = min(len, (int64_t)p.seg) + tid; j < p.seg; j += nt) {
            if (WHICH != 1) oi0[j] = -1;
            if (WHICH != 0) oi1[j] = -1;
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
        const int64_t id = p.ids[lo + j];
        if (!p.tau) {
            if (j < p.seg) {
                if (WHICH != 1) { ov0[j] = bias0 + s.x; oi0[j] = id; }
                if (WHICH != 0) { ov1[j] = bias1 + s.y; oi1[j] = id; }
            }
        } else {
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

}  // namespace

int ivfpq_scan(const gnnlm_ivfpq_scan_t& d, hipStream_t stream) {
    GNNLM_REQUIRE(d.n_tasks >= 0 && d.n_tasks < (1ll << 31), "ivfpq_scan: bad task count");
    if (d.n_tasks == 0) return OK;
    GNNLM_REQUIRE(d.codes && d.ids && d.list_off && d.lut && d.probe_list && d.probe_bias && d.task_q && d.task_p,
                  "ivfpq_scan: null operand");
    GNNLM_REQUIRE(d.M > 0 && d.M % 16 == 0 && d.M <= 128 && d.ld_lut >= (int64_t)d.M * 256 && d.ld_lut % 4 == 0 &&
                      (uintptr_t)d.lut % 16 == 0 && (uintptr_t)d.codes % 16 == 0,
                  "ivfpq_scan: need M % 16 == 0, M <= 128, 16-byte aligned tables");
    if (d.tau) GNNLM_REQUIRE(d.cand_val && d.cand_id && d.cand_cnt && d.cap > 0, "ivfpq_scan: filtered mode needs the candidate buffers");
    else GNNLM_REQUIRE(d.out_val && d.out_id && d.seg > 0 && d.ld_out >= d.seg, "ivfpq_scan: dense mode needs the output rows");
    const size_t lds = (size_t)d.M * 256 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        GNNLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ivfpq_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_set = true;
    }
    ProfScope prof(K_IVF, stream, 0.0, 0.0);
    if (d.M <= 64 && !getenv("GNNLM_IVF_SINGLE")) {
        static bool attr2_set = false;
        if (!attr2_set) {
            GNNLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ivfpq_scan2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
            attr2_set = true;
        }
        hipLaunchKernelGGL(ivfpq_scan2_kernel, dim3((unsigned)cdiv(d.n_tasks, 2)), dim3(1024), 2 * lds, stream, d);
        GNNLM_LAUNCH_CHECK();
        return OK;
    }
    hipLaunchKernelGGL(ivfpq_scan_kernel, dim3((unsigned)d.n_tasks), dim3(256), lds, stream, d);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

}  // namespace gnnlm
