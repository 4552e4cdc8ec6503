// IVF-PQ list scan on the int8 matrix cores, M = 64 (the reference's kNN index `OPQ64_1024,IVF4096,PQ64`, nprobe 32:
// gnnlm_scripts/wiki103/find_knn.sh:8-13, searched by faiss on the CPU at knn/knn_model.py:100).
//
// The f32 scan of ivfpq.hip does one 8-byte LDS read per (two queries, key, sub-quantizer) and one packed add: it runs at
// the LDS read rate with 4 bytes per (query, key, sub-quantizer).  Here the scan is a FILTER followed by an exact
// re-score of what passes, and the filter needs one BYTE per (query, key, sub-quantizer):
//
//   * a query's ADC table is quantised to 8 bits with a guaranteed one-sided bound (quantize_lut_kernel):
//         u[m][c] = min(255, floor((L[m][c] - lo_m) * inv)),   lo_m = min_c L[m][c],   L[m][c] < lo_m + (u + 1) delta
//     (stored as the signed byte u - 128: v_mfma_i32_*_i8 reads signed operands, the threshold absorbs the 128 * 64)
//     so that   score(q, x) = bias + sum_m L[m][code_m(x)]  <  bias + sum_lo + (sum_m u + M) delta + eps  =: UB(x);
//   * EIGHT queries that probe the same list share a workgroup; the LDS table entry of (sub-quantizer, code) is the 8
//     queries' bytes, so one ds_read_b64 serves 8 (query, key) pairs (LDS: 128 KiB of tables);
//   * the sums over the 64 sub-quantizers are taken by the matrix core: a lane's four look-ups ARE the 32-byte dense operand
//     of v_smfmac_i32_16x16x128_i8 (column = key, k = (look-up, query)), the other operand is the constant selector
//     S[query m][(look-up, query')] = [query' == m] -- two non-zeros in every four along k at most, i.e. exactly what the
//     2:4-SPARSE operand format holds: one instruction adds 16 keys x 16 sub-quantizers x 8 queries in the 16 busy cycles
//     the dense v_mfma_i32_16x16x64_i8 of round 3 took for half of that (PMC: SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA = 16.0
//     for both).  The integer sums are exact, the matrix core is the adder;
//   * a key survives for query j iff its integer sum reaches T_j = the integer image of the query's threshold tau (its
//     k-th best exact score after the dense round): UB(x) <= tau  =>  score(x) <= tau, so no key of the exact
//     one-pass scan (score > tau) is lost; survivors (row, list) are staged in LDS and appended to the query's list;
//   * ivfpq_rescore_kernel then recomputes the survivors' scores in float32 IN THE SUMMATION ORDER OF THE f32 SCAN
//     (ivfpq.hip, scan_rot), keeps score > tau and emits (score, payload[row]): the candidate set and every candidate's
//     bits are those of the one-pass f32 scan -- the search result is identical, by construction and by test.
//
// Look-ups are bank-conflict free by the same rotation idea as ivfpq.hip: a tile is 16 keys; lane (g = lane / 16,
// i = lane % 16) owns sub-quantizers 16 g .. 16 g + 15 of key i and reads them in the order 16 g + (i + p) % 16, the table
// is stored [half][code][32 slots] x 8 B, so the 32 lanes of an LDS access group hit 32 different slots; the key bytes are
// stored in that rotated order (gnnlm_ivfpq_pack_tiles), and the LDS address of a look-up is one v_perm_b32.
#include "kernels.h"

namespace gnnlm {
namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr int QG = 8;                                   // queries per workgroup
constexpr int TAB_BYTES = 2 * 256 * 32 * QG;            // [half][code][slot][query]: 128 KiB
#ifndef GNNLM_IVF8_WAVE_CAP
#define GNNLM_IVF8_WAVE_CAP 240
#endif
constexpr int WAVE_CAP = GNNLM_IVF8_WAVE_CAP;           // KEY entries (8 bytes: a key of the tile with its four queries' excesses) a wave stages per task
constexpr int SCAN_CTL = 768;                           // behind the tables: [16 waves][8] flush counters, then {counters, thresholds, queries}[8] of the group, twice (groups alternate)
constexpr int SCAN_LDS = TAB_BYTES + SCAN_CTL + 16 * WAVE_CAP * 8;
static_assert(SCAN_LDS <= 160 * 1024, "the filter's tables + staging regions exceed a CU's LDS");
constexpr int HIST_BINS = 1024, HIST_SHIFT = 4;         // threshold pass: sum_u (0 .. 16320) >> 4
constexpr int SUMS_LDS = TAB_BYTES + QG * HIST_BINS * 4;  // = 160 KiB: the whole LDS of a CU
constexpr int SURV_CNT_STRIDE = 16;                     // survivor counters one per 64-byte line: they are hammered by atomics
constexpr int QLUT_BYTES = 64 * 256;                    // one query's quantised table
// a survivor record is {row, list | sum_u << 18}: the key's integer sum (14 bits; SURV_SUM_BIG = "at least threshold + 254": the
// wave's LDS staging entry has one byte per query for the distance to the threshold) above an 18-bit list
constexpr int SURV_ROW_BITS = 19, SURV_LIST_BITS = 18, SURV_EXCESS_MAX = 254, SURV_SUM_BIG = 16383;

// codes [N, 64] row-major -> tiles of 16 rows, [tile][g 0..3][i 0..15][p 0..15] = code[16 tile + i][16 g + (i + p) % 16];
// rows beyond N are zero.  One thread per (row, g).
__global__ __launch_bounds__(256) void pack_tiles_kernel(const uint8_t* __restrict__ codes, int64_t N, uint8_t* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = e >> 2;
    const int g = (int)(e & 3);
    if (row >= ((N + 15) >> 4 << 4)) return;
    uint32_t v[4] = {0u, 0u, 0u, 0u};
    if (row < N) {
        const uint8_t* src = codes + row * 64 + 16 * g;
        const int i = (int)(row & 15);
#pragma unroll
        for (int p = 0; p < 16; ++p) v[p >> 2] |= (uint32_t)src[(i + p) & 15] << (8 * (p & 3));
    }
    *reinterpret_cast<uint4*>(out + (((row >> 4) * 4 + g) * 16 + (row & 15)) * 16) = uint4{v[0], v[1], v[2], v[3]};
}

// One workgroup per query: L [64][256] f32 -> u8 [half][code][32 slots] + {delta, sum_lo, absmax, 0}.
__global__ __launch_bounds__(256) void quantize_lut_kernel(const float* __restrict__ lut, int64_t ld, uint8_t* __restrict__ qlut,
                                                           float* __restrict__ qmeta) {
    __shared__ __attribute__((aligned(16))) float tab[64 * 256];
    __shared__ float lo_s[64], rng_s[64], amax_s[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t q = blockIdx.x;
    {
        const float4* src = reinterpret_cast<const float4*>(lut + q * ld);
        float4* dst = reinterpret_cast<float4*>(tab);
        for (int e = tid; e < 64 * 64; e += 256) dst[e] = src[e];
    }
    __syncthreads();
    for (int m = wave; m < 64; m += 4) {
        float mn = INFINITY, mx = -INFINITY;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const float v = tab[m * 256 + lane + 64 * x];
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
        }
        mn = -wave_max(-mn);
        mx = wave_max(mx);
        if (lane == 0) { lo_s[m] = mn; rng_s[m] = mx - mn; amax_s[m] = fmaxf(fabsf(mn), fabsf(mx)); }
    }
    __syncthreads();
    float maxrange = 0.f, sum_lo = 0.f, amax = 0.f;
    for (int m = 0; m < 64; ++m) {                       // every thread, same order: sum_lo is one fixed f32 chain
        maxrange = fmaxf(maxrange, rng_s[m]);
        sum_lo += lo_s[m];
        amax = fmaxf(amax, amax_s[m]);
    }
    // inv a little below 255 / maxrange, delta a little above maxrange / 255: (u + 1) delta bounds L - lo from above whatever
    // the rounding of the two float operations of the quantisation did (see the header comment; delta * inv >= 1 + 2^-19)
    const bool flat = !(maxrange > 0.f);
    const float inv = flat ? 0.f : (255.f / maxrange) * (1.f - 3.8146973e-6f);          // 1 - 2^-18
    const float delta = flat ? 1e-30f : (maxrange / 255.f) * (1.f + 7.6293945e-6f);     // 1 + 2^-17
    if (tid == 0) {
        qmeta[q * 4 + 0] = delta;
        qmeta[q * 4 + 1] = sum_lo;
        qmeta[q * 4 + 2] = amax;
        qmeta[q * 4 + 3] = 0.f;
    }
    uint8_t* dst = qlut + q * QLUT_BYTES;
    for (int hc = tid; hc < 512; hc += 256) {           // row (half, code): 32 slots = 32 bytes
        const int h = hc >> 8, c = hc & 255;
        uint32_t o[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            uint32_t pk = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int m = 32 * h + 4 * w + b;
                const float y = (tab[m * 256 + c] - lo_s[m]) * inv;
                const int u = min(255, max(0, (int)floorf(y)));
                pk |= (uint32_t)(u ^ 0x80) << (8 * b);            // stored as the SIGNED byte u - 128: the i8 MFMA reads signed operands
            }
            o[w] = pk;
        }
        uint4* d4 = reinterpret_cast<uint4*>(dst + hc * 32);
        d4[0] = uint4{o[0], o[1], o[2], o[3]};
        d4[1] = uint4{o[4], o[5], o[6], o[7]};
    }
}

// The integer image of a query's threshold for one list: a key can score above tau only if the MFMA's sum (sum_u - 128 * 64: the
// table bytes are u - 128) reaches it.  eps: rounding of the f32 score chain (65 adds), of sum_lo and of this formula's
// subtractions: (M + 2) 2^-22 sum |terms|.  survive iff sum_u + 64 > thr' with thr' in (thr - 1, thr + 1)
__device__ __forceinline__ int filter_threshold(const float* __restrict__ qmeta, int64_t q, float bias, float tau) {
    const float delta = qmeta[q * 4], sum_lo = qmeta[q * 4 + 1], amax = qmeta[q * 4 + 2];
    const float eps = 66.f * 2.3841858e-7f * (64.f * amax + fabsf(bias) + (fabsf(tau) < INFINITY ? fabsf(tau) : 0.f));
    const float thr = ((tau - bias) - sum_lo - eps) / delta - 64.f;
    return !(thr == thr) || thr <= -1.0e9f ? -(1 << 30) : (thr >= 1.0e9f ? 0x7fffffff : (int)floorf(thr) - 128 * 64);
}

// ---- the scan's task table (gnnlm_ivfpq_build_groups): pairs per list, group offsets, scatter.  cnt / fill: [nlist + 1] each
__global__ __launch_bounds__(256) void groups_fill_kernel(int32_t* grp_list, int32_t* grp_q, int64_t* grp_out, int64_t G, int32_t* scratch, int nlist) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < G) grp_list[e] = -1;
    if (e < G * QG) { grp_q[e] = -1; if (grp_out) grp_out[e] = -1; }
    if (e < 2 * (int64_t)(nlist + 1)) scratch[e] = 0;
}
__global__ __launch_bounds__(256) void groups_count_kernel(const int64_t* __restrict__ pl, int64_t ld, int64_t n, int P, int nlist, int32_t* cnt) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * P) return;
    const int64_t l = pl[(e / P) * ld + e % P];
    if (l >= 0 && l < nlist) atomicAdd(&cnt[l], 1);
}
// one workgroup: cnt[l] -> the first group of list l (exclusive prefix of ceil(cnt / 8)), in place; n_groups = the total
__global__ __launch_bounds__(1024) void groups_scan_kernel(int32_t* cnt, int nlist, int32_t* n_groups) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (nlist + 1023) / 1024, lo = tid * per, hi = min(nlist, lo + per);
    int mine = 0;
    for (int l = lo; l < hi; ++l) mine += (cnt[l] + QG - 1) / QG;
    part[tid] = mine;
    __syncthreads();
    for (int step = 1; step < 1024; step <<= 1) {                            // inclusive prefix over the threads
        const int other = tid >= step ? part[tid - step] : 0;
        __syncthreads();
        part[tid] += other;
        __syncthreads();
    }
    int at = part[tid] - mine;
    for (int l = lo; l < hi; ++l) { const int g = (cnt[l] + QG - 1) / QG; cnt[l] = at; at += g; }
    if (tid == 1023) *n_groups = part[1023];
}
__global__ __launch_bounds__(256) void groups_scatter_kernel(const int64_t* __restrict__ pl, int64_t ld, int64_t n, int P, int nlist, int64_t seg,
                                                             const int32_t* __restrict__ goff, int32_t* fill, int32_t* grp_list, int32_t* grp_q,
                                                             int64_t* grp_out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * P) return;
    const int64_t q = e / P, l = pl[q * ld + e % P];
    if (l < 0 || l >= nlist) return;
    const int pos = atomicAdd(&fill[l], 1);
    const int64_t g = goff[l] + pos / QG;
    grp_list[g] = (int32_t)l;
    grp_q[g * QG + pos % QG] = (int32_t)q;
    if (grp_out) grp_out[g * QG + pos % QG] = e * seg;
}

#define GNNLM_PERM(hi_, lo_, sel_) __builtin_amdgcn_perm((hi_), (lo_), (sel_))
#ifndef GNNLM_IVF8_NW
#define GNNLM_IVF8_NW 16         // waves per workgroup of the scan (A/B: 8)
#endif
constexpr int NW = GNNLM_IVF8_NW, NTH = 64 * NW;
#ifndef GNNLM_IVF8_PF
#define GNNLM_IVF8_PF 3         // tiles of code bytes in flight per wave of the filter = steps per block of the loop (every register name is static).
                                // A/B with exact waits, medians of 7 searches: 3: 8.17 ms, 4: 9.06, 6: 8.82, 8: 9.21 -- depth buys nothing, the steps past
                                // the end of a list (up to PF - 1 per wave and group) cost
#endif
#ifndef GNNLM_IVF8_PF_SUMS
#define GNNLM_IVF8_PF_SUMS 3    // ... of the threshold pass
#endif
#ifndef GNNLM_IVF8_EXP
#define GNNLM_IVF8_EXP 0        // ablation / instrumented builds: 1 no code loads, 2 no table fill, 4 no look-ups, 8 no matrix instructions, 32 nothing survives,
#endif                          // 64 no staging writes, 128 the staging count stays 0, 256 always the same four tiles; 512 phase times, 1024 the waves' exit times
                                // (into the spare words of work_ctr: tools/ivf8_phases.py, tools/ivf8_waves.py)

// SUMS = false: the filter (survivors of the integer threshold).  SUMS = true: the threshold pass -- the integer sums sum_u of a
// list's keys are HISTOGRAMMED per query in LDS (bins of 16, atomics without return) and the (query, list) histogram is written
// to the segment grp_out names; nothing is compared, no per-key output.
template <bool SUMS>
__global__ __launch_bounds__(NTH) void ivfpq_scan8_kernel(gnnlm_ivfpq_scan8_t p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // tables at LDS address 0 (look-up addresses are absolute)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // consecutive workgroups go to consecutive XCDs: give each XCD a contiguous range of the list-sorted groups
    // The workgroups are PERSISTENT (one per CU: the tables take the CU's LDS): workgroup b of XCD b % 8 walks the groups b / 8,
    // b / 8 + gridDim / 8, ... of its XCD's range -- the groups of one list are neighbours in that order, so they run side by
    // side on one XCD and share the list's bytes in its L2; no workgroup launch (and LDS allocation) between two groups.
    const int n_groups = min(*p.n_groups, p.max_groups);
    const int per_xcd = (n_groups + 7) >> 3;
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem != 0u) __builtin_trap();
    // The first group of a workgroup is b / 8; the next ones come from the XCD's counter (work_ctr: lists of different lengths
    // stay balanced) or, without one, by striding
    bool first = true;
    int gi = (int)(blockIdx.x >> 3);
    int* next_s = reinterpret_cast<int*>(smem + TAB_BYTES);                   // (free between two groups)
#if GNNLM_IVF8_EXP & 512
    // instrumented build: thread 0's time per phase (group set-up + table fill | tile loop | end of the group) into the spare words of work_ctr
    long long stamp = clock64();
    const long long k0 = stamp, w0 = wall_clock64();
    auto phase = [&](int ph) { const long long t = clock64(); if (tid == 0 && p.work_ctr) atomicAdd(&p.work_ctr[(blockIdx.x & 7) * 16 + 1 + ph], (int)(t - stamp)); stamp = t; };
    // ... and inside the steps of wave 8 (the third of its SIMD): accumulated in registers, added at the end of the group
    long long st2 = 0; int acc_t[4] = {0, 0, 0, 0};
    auto tick = [&](int ph) { if (wave == 8) { const long long t = clock64(); if (ph >= 0) acc_t[ph] += (int)(t - st2); st2 = t; } };
#else
    auto phase = [&](int) {};
    auto tick = [&](int) {};
#endif
    int nxt = 0;                                                             // (thread 0) the counter's value for the group after this one
    int gi_next = 0, par = 0;
    // the index of the next group: thread 0's to everybody (three barriers; the filter's groups do this inside the two barriers their end has anyway)
    auto next_index = [&]() -> int { return p.work_ctr ? (int)(gridDim.x >> 3) + nxt : gi + (int)(gridDim.x >> 3); };
    auto advance_slow = [&]() {
        __syncthreads();                                                     // the group's tables and histograms are done with
        if (tid == 0) *next_s = next_index();
        __syncthreads();
        gi_next = *next_s;
        __syncthreads();
    };
    for (;;) {
    phase(2);
    if (!first) gi = gi_next;
    first = false;
    if (gi >= per_xcd) break;
    const int grp = (int)(blockIdx.x & 7) * per_xcd + gi;
    if (grp >= n_groups) break;
    // the NEXT group's index is asked for now and read at the end of this group: the atomic's round trip (and nothing but it) used to
    // stand between two groups, 135 times per workgroup
    if (tid == 0 && p.work_ctr) nxt = atomicAdd(&p.work_ctr[(blockIdx.x & 7) * 16], 1);
    const int list = p.grp_list[grp];
    if (list < 0) { advance_slow(); continue; }
    const int64_t lo = p.list_off[list], hi = p.list_off[list + 1];
    if (hi <= lo) {
        // an empty list: the threshold pass still owes ivfpq_tau_kernel a (zero) histogram for every (query, list) pair of the group
        if (SUMS) {
            for (int u = 0; u < QG; ++u) {
                const int64_t ob = p.grp_out[(int64_t)grp * QG + u];
                if (p.grp_q[(int64_t)grp * QG + u] < 0 || ob < 0) continue;
                for (int e = tid; e < HIST_BINS; e += NTH) p.out_hist[ob + e] = 0u;
            }
        }
        advance_slow();
        continue;
    }
    // this wave's staged KEY entries: {(row - lo) << 3 | 4 (g & 1), the four queries' bytes min(255, excess + 1) (0: no survivor)}
    uint2* wbuf = reinterpret_cast<uint2*>(smem + TAB_BYTES + SCAN_CTL) + __builtin_amdgcn_readfirstlane(wave) * WAVE_CAP;   // (a wave-uniform base: scalar)
    int* wc = reinterpret_cast<int*>(smem + TAB_BYTES) + wave * QG;          // this wave's per-slot counters / first positions (a flush in the middle of a group)
    // the group's [0..8) counters / first positions, [8..16) thresholds, [16..24) queries; two copies in turn: a wave that is done with a group
    // sets the next one up while others still write the records of this one
    par ^= 1;
    int* wgc = reinterpret_cast<int*>(smem + TAB_BYTES + 512) + 32 * par;
    const int* gq = p.grp_q + (int64_t)grp * QG;
    uint2* surv = reinterpret_cast<uint2*>(p.surv);                           // {row, list} per survivor
    int qs[QG];
#pragma unroll
    for (int j = 0; j < QG; ++j) qs[j] = gq[j];

    // ---- the 8 queries' byte tables -> [half][code][slot] x 8 B: a 4 x 8 byte transpose per thread and step
#if !(GNNLM_IVF8_EXP & 2)
    {
        constexpr int FILL = QLUT_BYTES / 4 / NTH;       // dwords of each query's table per thread
        constexpr int FB = SUMS ? 1 : FILL;              // of which in flight at once (the threshold pass has no registers to spare)
#pragma unroll
        for (int i0 = 0; i0 < FILL; i0 += FB) {
            uint32_t w[FB][QG];
#pragma unroll
            for (int it = 0; it < FB; ++it)
#pragma unroll
                for (int j = 0; j < QG; ++j)
                    w[it][j] = qs[j] >= 0 ? reinterpret_cast<const uint32_t*>(p.qlut + (int64_t)qs[j] * QLUT_BYTES)[tid + (i0 + it) * NTH] : 0u;
#pragma unroll
            for (int it = 0; it < FB; ++it) {
                uint32_t o[8];
#pragma unroll
                for (int hq = 0; hq < 2; ++hq) {         // queries 4 hq .. 4 hq + 3 -> dword hq of the four entries
                    const uint32_t a = w[it][4 * hq], b = w[it][4 * hq + 1], c = w[it][4 * hq + 2], d = w[it][4 * hq + 3];
                    const uint32_t t0 = GNNLM_PERM(b, a, 0x05010400u), t1 = GNNLM_PERM(b, a, 0x07030602u);
                    const uint32_t t2 = GNNLM_PERM(d, c, 0x05010400u), t3 = GNNLM_PERM(d, c, 0x07030602u);
                    o[0 + hq] = GNNLM_PERM(t2, t0, 0x05040100u);
                    o[2 + hq] = GNNLM_PERM(t2, t0, 0x07060302u);
                    o[4 + hq] = GNNLM_PERM(t3, t1, 0x05040100u);
                    o[6 + hq] = GNNLM_PERM(t3, t1, 0x07060302u);
                }
                uint4* dst = reinterpret_cast<uint4*>(smem) + 2 * (tid + (i0 + it) * NTH);
                dst[0] = uint4{o[0], o[1], o[2], o[3]};
                dst[1] = uint4{o[4], o[5], o[6], o[7]};
            }
        }
    }
#endif
    const int j = lane & 15, g = lane >> 4;
    const int qs_lane = lane < QG ? gq[lane] : -1;                           // lanes 0..7: the query of slot `lane` (flush of the survivors)
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem + TAB_BYTES);          // SUMS: [8 queries][HIST_BINS] counters
    if (SUMS) for (int e = tid; e < QG * HIST_BINS; e += NTH) hist[e] = 0u;
    // ---- the D layout of the sparse instruction: D[query 4 g + r][key j] in register r of lane (g, j) -- lanes 0..31 hold the 8 queries
    // of the group for the 16 keys of a tile, lanes 32..63 the unused rows 8..15 of the selector.  Integer thresholds per register
    // (clamped to +-16384: a sum lies in [-8192, 8128], so the test is the same and sum - T cannot overflow)
    constexpr int T_NEVER = 16384;
    int T4[4] = {T_NEVER, T_NEVER, T_NEVER, T_NEVER};
    // SUMS: lane (g, j) counts two queries of key j: g < 2 its own registers 0, 1 (queries 4 g, 4 g + 1), g >= 2 the registers 2, 3 of
    // lane (g - 2, j), which arrive by v_permlane32_swap (queries 4 (g - 2) + 2, + 3): two atomics per lane on all 64 lanes
    // (odd keys count their two queries in the other order: every atomic instruction then spreads over all 8 histograms, 8 lanes each --
    // counters of one address serialise)
    uint32_t *hist_a = nullptr, *hist_b = nullptr;
    if (SUMS) {
        const int q0 = 4 * (g & 1) + (g >= 2 ? 2 : 0) + (j & 1), q1 = q0 ^ 1;
        if (gq[q0] >= 0 && p.grp_out[(int64_t)grp * QG + q0] >= 0) hist_a = hist + q0 * HIST_BINS;
        if (gq[q1] >= 0 && p.grp_out[(int64_t)grp * QG + q1] >= 0) hist_b = hist + q1 * HIST_BINS;
    } else if (g < 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qr = gq[4 * g + r];
            if (qr >= 0) T4[r] = max(-T_NEVER, min(T_NEVER, filter_threshold(p.qmeta, qr, p.coarse[(int64_t)qr * p.ld_coarse + list], p.tau[qr])));
        }
    }
#if GNNLM_IVF8_EXP & 32
    if (p.cap > 0) T4[0] = T4[1] = T4[2] = T4[3] = T_NEVER;          // nothing survives, but the compiler cannot know: every instruction stays
#elif GNNLM_IVF8_EXP & 255
    T4[0] = T4[1] = T4[2] = T4[3] = T_NEVER;                         // ablation builds time the main loop: nothing survives
#endif
    if (!SUMS && wave == 0) {
        int Tl = 0;                                                          // lane s < 8: the threshold of slot s
#pragma unroll
        for (int sl = 0; sl < QG; ++sl) {
            const int t = __builtin_amdgcn_readlane(T4[sl & 3], 16 * (sl >> 2));
            if (lane == sl) Tl = t;
        }
        if (lane < QG) { wgc[lane] = 0; wgc[8 + lane] = Tl; wgc[16 + lane] = qs_lane; }
    }
    // the accumulators start at 1 - T: register r then holds the key's EXCESS over query 4 g + r's threshold PLUS ONE, a survivor is a
    // positive one -- and v_sat_pk_u8_i16 turns the four registers into the entry's four bytes (0: not a survivor)
    const v4i negT = SUMS ? v4i{0, 0, 0, 0} : v4i{1 - T4[0], 1 - T4[1], 1 - T4[2], 1 - T4[3]};
    __syncthreads();                                                       // the tables are in place
    phase(0);
#if GNNLM_IVF8_EXP & 1024
    const long long g0 = clock64();
#endif

    // ---- look-up constants of the lane: slot byte offsets of look-ups 2 s / 2 s + 1 and the table half in byte 2
    uint32_t tc[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const uint32_t s0 = (uint32_t)(16 * (g & 1) + ((j + 2 * s) & 15)) << 3, s1 = (uint32_t)(16 * (g & 1) + ((j + 2 * s + 1) & 15)) << 3;
        tc[s] = s0 | s1 << 8 | (uint32_t)(g >> 1) << 16;
    }
    // The selector S[query m][(look-up, query')] = [query' == m] as the SPARSE operand of v_smfmac_i32_16x16x128_i8 (2 kept values out of every
    // 4 along k; layout measured with tools/probes/smfmac_layout.hip): lane (kb, m) holds 16 kept bytes -- bytes 8 h .. 8 h + 7 meet a 16-byte
    // stretch of the dense operand (two look-ups of 8 query bytes), kept byte s its group of four bytes 4 (s / 2) .. + 3 at the position the
    // index register names (2 bits per kept byte).  Row m < 8 keeps a 1 at query byte m of both look-ups of each stretch, rows 8 .. 15 nothing.
    uint32_t sel[4] = {0u, 0u, 0u, 0u}, sidx = 0xccccccccu;                  // pairs default to positions (0, 3)
    if (j < QG) {
        const int pos = j & 3;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int lk = 0; lk < 2; ++lk) {
                const int s0 = 8 * h + 2 * (2 * lk + (j >> 2));               // the kept pair of the group that holds byte m of look-up lk
                const int sb = pos < 3 ? s0 : s0 + 1;                        // (position 3 goes to the pair's second byte, whose index is 3 already)
                sel[sb >> 2] |= 1u << (8 * (sb & 3));
                if (pos < 3) sidx = (sidx & ~(3u << (2 * s0))) | (uint32_t)pos << (2 * s0);
            }
    }
    const v4i Asel = {(int)sel[0], (int)sel[1], (int)sel[2], (int)sel[3]};

    // ---- the list's tiles: wave w takes tiles w, w + 16, ... two per step.  Everything that steers the loop is scalar (the wave
    // index through readfirstlane), the code bytes come by buffer loads off one resource over the list's tile range (lane
    // offset in a VGPR, tile offset in an SGPR, out-of-range tiles read as zeros): no vector instruction is spent on addresses
    const int64_t t_lo = lo >> 4;
    const int nt = (int)(((hi - 1) >> 4) - t_lo) + 1;                        // tiles that hold rows of the list
    const int len = (int)(hi - lo), row_shift = (int)(lo & 15);               // local row of (tile u, key i) = 16 u + i - row_shift
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.tiles) + t_lo * 1024, 0, nt * 1024, 0x00020000);
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int voff = lane * 16;
    const uint32_t rs_lane = (uint32_t)((j - row_shift) * 8) + (uint32_t)(4 * (g & 1));   // the lane's part of an entry's row word: (row << 3 | slot base) - 128 u
    auto load_tile = [&](int u) -> v4u {
#if GNNLM_IVF8_EXP & 1
        return v4u{(uint32_t)u * 2654435761u + lane, (uint32_t)u * 40503u ^ lane, (uint32_t)u + 77u * lane, (uint32_t)u * 3u + lane};
#elif GNNLM_IVF8_EXP & 256
        return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, __builtin_amdgcn_readfirstlane((u & 3) * 1024), 0);   // always the same four tiles: cache hits
#else
        return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, __builtin_amdgcn_readfirstlane(u * 1024), 0);   // (u is wave-uniform: tell the compiler)
#endif
    };
    // look-ups 4 q .. 4 q + 3 of one tile (= the 32-byte dense operand of one matrix instruction): address = {0, half, code byte, slot
    // offset} by one v_perm_b32 each
    auto lookups4 = [&](const v4u& cw, int q, v8i& X) __attribute__((always_inline)) {
        const uint32_t w[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int pp = 4 * q + e;                                        // look-up pp: byte pp of the lane's 16 code bytes
            const uint32_t ad = GNNLM_PERM(w[pp >> 2], tc[pp >> 1], 0x0c020000u | (uint32_t)(4 + (pp & 3)) << 8 | (uint32_t)(pp & 1));
#if GNNLM_IVF8_EXP & 4
            const u32x2 x = u32x2{ad, ad ^ 0x55u};
#else
            const u32x2 x = *reinterpret_cast<const __attribute__((address_space(3))) u32x2*>(ad);
#endif
            X[2 * e] = (int)x.x;
            X[2 * e + 1] = (int)x.y;
        }
    };
    // Survivors of one tile.  A wave stages KEY entries in its OWN LDS region: a lane whose key survives for at least one of its
    // four queries writes {local row, slot base; one byte per query: min(255, excess + 1), 0 = no survivor} at a position the wave
    // computes itself: a scalar count of what it has staged so far + the lane's rank inside the tile's ONE compare mask (v_mbcnt).
    // Straight-line code for every tile (round 5 appended per query register, each behind its own scalar test and branch: the
    // compare / append phase of a tile took as long as its look-ups, tools/ivf8_phases.py); no LDS atomic, nothing to wait for.
    // A full region is flushed by the wave alone, the regions left at the end of the group by the workgroup together: one global
    // atomic per query slot for the first positions (below).
    int wcnt = 0;                                                            // (scalar) entries staged by this wave
    // A staged region -> the queries' lists, in two passes over the region (it stays in LDS; nothing is carried in registers between
    // them -- round 5 kept every entry and its rank in registers across the end-of-group barriers, 10 of the kernel's 128):
    // (1) every (entry, query) survivor adds one to its query slot's counter, an LDS atomic WITHOUT return on one of 8 counters (the
    // wave's own for a flush in the middle of a group, the workgroup's at its end); lanes 0..7 turn the totals into first positions
    // with ONE global atomic per slot and leave them in the counters; (2) every survivor takes its position from its slot's counter
    // (LDS atomic with return) and goes there.
    auto count_entries = [&](int* ctr) __attribute__((always_inline)) {
        for (int c = lane; c < wcnt; c += 64) {
            const uint2 e = wbuf[c];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (e.y >> (8 * r) & 255u) atomicAdd(&ctr[(e.x & 4u) + r], 1);
        }
    };
    auto write_entries = [&](int* pos) __attribute__((always_inline)) {
        for (int c = lane; c < wcnt; c += 64) {
            const uint2 e = wbuf[c];
            const uint32_t row = (uint32_t)(lo + ((e.x >> 3) & ((1u << SURV_ROW_BITS) - 1u)));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b8 = (int)(e.y >> (8 * r) & 255u);
                if (b8 == 0) continue;
                const int sl = (int)(e.x & 4u) + r;
                const int64_t at = atomicAdd(&pos[sl], 1);
                const int Tsl = wgc[8 + sl], qsl = wgc[16 + sl];
                if (at < p.cap && qsl >= 0) {
                    // the key's integer sum sum_u = excess + T + 128 * 64 (0 .. 16320); SURV_SUM_BIG: more than the entry's byte could hold
                    const int ex = b8 - 1;
                    const int su = ex >= SURV_EXCESS_MAX ? SURV_SUM_BIG : min(SURV_SUM_BIG - 1, max(0, ex + Tsl + 128 * 64));
                    surv[(int64_t)qsl * p.cap + at] = uint2{row, (uint32_t)list | (uint32_t)su << SURV_LIST_BITS};
                }
            }
        }
        wcnt = 0;
    };
    // a full region in the middle of a group (the best lists of a query hold thousands of its survivors): the wave flushes alone
    auto flush_wave = [&]() __attribute__((always_inline)) {
        if (wcnt == 0) return;
        if (lane < QG) wc[lane] = 0;
        count_entries(wc);
        if (lane < QG) {
            const int tot = wc[lane];
            wc[lane] = (tot > 0 && qs_lane >= 0) ? atomicAdd(&p.surv_cnt[(int64_t)qs_lane * SURV_CNT_STRIDE], tot) : 0;
        }
        write_entries(wc);
    };
    // Software pipeline at the grain of ONE matrix instruction: the four look-ups behind instruction q of tile i + 1 are issued right after
    // instruction q of tile i has read the same registers (one set of 32 look-up registers per lane; a second set, the next tile's look-ups
    // all issued ahead of this tile's instructions, measured the same: 8.35 against 8.17 ms).  The code bytes of tile i + PF are requested at
    // the top of step i into the registers tile i's bytes leave; the loop is unrolled PF times so that every name is static.
    constexpr int PF = SUMS ? GNNLM_IVF8_PF_SUMS : GNNLM_IVF8_PF;
    v8i X[4];
    // (the threshold pass may histogram a SAMPLE of the list: every TS-th tile, starting at a tile that differs from list to list)
    const int TS = SUMS ? max(1, p.sums_stride) : 1;
    auto step = [&](const v4u& cn, v4u& cl, int u) __attribute__((always_inline)) {
        tick(-1);
        cl = load_tile(u + PF * NW * TS);
        v4i acc = negT;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#if GNNLM_IVF8_EXP & 8
            acc += v4i{X[q][0], X[q][1], X[q][2], X[q][3]} + v4i{X[q][4], X[q][5], X[q][6], X[q][7]};
#else
            acc = __builtin_amdgcn_smfmac_i32_16x16x128_i8(Asel, X[q], acc, (int)sidx, 0, 0);
#endif
            lookups4(cn, q, X[q]);
            // nothing crosses: left alone, the scheduler deals the 16 reads out look-up-major (by address register), every quarter is then
            // complete only when nearly all 16 reads are, and the first matrix instruction of the next step waits for a drained queue
            __builtin_amdgcn_sched_barrier(0);
        }
        tick(0);
        const int row = 16 * u + j - row_shift;                              // the lane's key (local row of the list)
        const bool edge = u == 0 || u >= nt - 1;                             // the two edge tiles share rows with the neighbouring lists (and beyond: nothing)
        if (SUMS) {
            // (LDS atomics without return: nothing waits for them)
            const gnnlm_u32x2 w2 = __builtin_amdgcn_permlane32_swap((uint32_t)acc[2], (uint32_t)acc[2], false, false);
            const gnnlm_u32x2 w3 = __builtin_amdgcn_permlane32_swap((uint32_t)acc[3], (uint32_t)acc[3], false, false);
            const int e0 = g >= 2 ? (int)w2.x : acc[0], e1 = g >= 2 ? (int)w3.x : acc[1];
            const int s0 = (j & 1) ? e1 : e0, s1 = (j & 1) ? e0 : e1;
            if (!edge || (unsigned)row < (unsigned)len) {
                if (hist_a) atomicAdd(&hist_a[(s0 + 128 * 64) >> HIST_SHIFT], 1u);
                if (hist_b) atomicAdd(&hist_b[(s1 + 128 * 64) >> HIST_SHIFT], 1u);
            }
        } else {
            // the entry's four bytes: registers 0, 1 and 2, 3 packed as int16 pairs (one v_perm_b32 each; |value| < 2^15), saturated to
            // unsigned bytes (v_sat_pk_u8_i16: <= 0 -> 0, > 255 -> 255), one compare for the whole lane, one LDS write
            if (wcnt + 32 > WAVE_CAP) flush_wave();                          // a tile adds at most 32 entries (lanes 0 .. 31: 2 x 16 keys' query halves)
            tick(1);
            uint32_t lo8, hi8;
            const uint32_t p01 = GNNLM_PERM((uint32_t)acc[1], (uint32_t)acc[0], 0x05040100u), p23 = GNNLM_PERM((uint32_t)acc[3], (uint32_t)acc[2], 0x05040100u);
            asm("v_sat_pk_u8_i16 %0, %1" : "=v"(lo8) : "v"(p01));
            asm("v_sat_pk_u8_i16 %0, %1" : "=v"(hi8) : "v"(p23));
            const uint32_t ex4 = GNNLM_PERM(hi8, lo8, 0x05040100u);             // (the low halves of both: whatever the instruction leaves in the upper ones)
            // rows outside the list (the edge tiles, steps past the list's end) fail ONE unsigned compare of the entry's row word against a
            // scalar limit: (row << 3 | slot) < lim, lim = len << 3 on an edge tile, everything elsewhere (a negative row is a huge word)
            const uint32_t rowslot = ((uint32_t)(16 * u) << 3) + rs_lane;
            const uint32_t lim = edge ? (uint32_t)len << 3 : 0xffffffffu;
            uint32_t ex4r = rowslot < lim ? ex4 : 0u;
            asm("" : "+v"(ex4r));                                            // (opaque: left alone, the compiler turns the select back into an AND of two compares and
            const bool keep = ex4r != 0u;                                    //  re-materialises the mask through a VGPR for the ballot)
            const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
            const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (keep) wbuf[wcnt + rank] = uint2{rowslot, ex4r};
            wcnt = __builtin_amdgcn_readfirstlane(wcnt + __builtin_popcountll(m));   // (scalar: the compiler's divergence analysis gives up on it)
            tick(2);
        }
    };
    {
        int u = SUMS ? wv * TS + list % TS : wv;                                  // (by list, not by group: which queries share a group is not specified)
        v4u C[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) C[i] = load_tile(u + i * NW * TS);        // (loads beyond the list's tiles return zeros)
#pragma unroll
        for (int q = 0; q < 4; ++q) lookups4(C[0], q, X[q]);
        // The loop is left only between two blocks of PF steps and every step of a block RUNS -- a step beyond the list's tiles reads zeros (the
        // buffer's range check) and takes the edge tiles' path, where its rows fail the range test.  With an exit or a skipped step inside the
        // block the compiler's s_waitcnt insertion merges paths with different loads in flight and falls back to waiting for (nearly) all of
        // them: `vmcnt(1)` with seven tiles in flight, once per block -- the waves stood at the code loads 17 % of their time (PMC).
#define GNNLM_IVF8_STEP(t) if constexpr (PF > (t)) { step(C[((t) + 1) % PF], C[(t)], u); u += NW * TS; }
        while (u < nt) {
            GNNLM_IVF8_STEP(0) GNNLM_IVF8_STEP(1) GNNLM_IVF8_STEP(2) GNNLM_IVF8_STEP(3)
            GNNLM_IVF8_STEP(4) GNNLM_IVF8_STEP(5) GNNLM_IVF8_STEP(6) GNNLM_IVF8_STEP(7)
        }
#undef GNNLM_IVF8_STEP
        static_assert(PF >= 3 && PF <= 8, "GNNLM_IVF8_PF");
        phase(1);
#if GNNLM_IVF8_EXP & 512
        if (tid == 512 && p.work_ctr) for (int i = 0; i < 3; ++i) { atomicAdd(&p.work_ctr[(blockIdx.x & 7) * 16 + 8 + i], acc_t[i]); acc_t[i] = 0; }
        if (tid == 512 && p.work_ctr) atomicAdd(&p.work_ctr[(blockIdx.x & 7) * 16 + 11], (nt - wv + NW - 1) / NW);
#endif
#if GNNLM_IVF8_EXP & 1024
        // when does each wave leave the tile loop?  (ticks since the group's tables were in place; slot 15 <- waves 0 and 15)
        if (lane == 0 && p.work_ctr) atomicAdd(&p.work_ctr[(blockIdx.x & 7) * 16 + (wave ? wave : 15)], (int)((clock64() - g0) >> 4));
#endif
    }
    if (SUMS) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < QG; ++u) {
            const int64_t ob = p.grp_out[(int64_t)grp * QG + u];
            if (qs[u] < 0 || ob < 0) continue;
            for (int e = tid; e < HIST_BINS; e += NTH) p.out_hist[ob + e] = hist[u * HIST_BINS + e];
        }
        advance_slow();
        continue;
    }
    // ---- end of the group: the 16 waves' regions -> the queries' lists with ONE global atomic per query slot for the whole
    // workgroup (the counters are contended: 30 lists x their groups add to every query's): the entries take their ranks on the
    // workgroup's 8 LDS counters (zeroed when the group was set up)
    count_entries(wgc);
    if (tid == 0) *next_s = next_index();                                    // (the next group's index rides on the same two barriers)
    phase(3);
    __syncthreads();
    phase(4);
    if (tid < QG) {
        const int tot = wgc[tid];
        wgc[tid] = (tot > 0 && qs_lane >= 0) ? atomicAdd(&p.surv_cnt[(int64_t)qs_lane * SURV_CNT_STRIDE], tot) : 0;
    }
    __syncthreads();
    gi_next = *next_s;
    phase(5);
    write_entries(wgc);
    // no barrier here: the tables are free (every wave has left its tile loop), the staging regions are the waves' own, the next group's
    // counters are the other copy, and nobody flushes (the wave-0 counters hold *next_s) before the next "tables in place" barrier
    }
#if GNNLM_IVF8_EXP & 512
    if (tid == 0 && p.work_ctr) {
        atomicAdd(&p.work_ctr[(blockIdx.x & 7) * 16 + 12], (int)((clock64() - k0) >> 4));
        atomicAdd(&p.work_ctr[(blockIdx.x & 7) * 16 + 13], (int)(wall_clock64() - w0));
    }
#endif
}

// Threshold from the histograms of a query's D dense lists (written by the SUMS pass): a LOWER bound of its k-th best exact
// score.  score(x) >= bias_l + sum_lo + sum_u(x) delta' (delta' a hair below delta), so with v(x) = sum_u(x) + off_l,
// off_l <= (bias_l - bias_min) / delta, the k-th largest v gives tau = bias_min + sum_lo + v_k delta' - eps: at least k keys score
// >= tau.  Bins of 16: a key of bin b of list l counts in the combined bin b + (off_l >> 4) -- its lower edge bounds v from below.
__global__ __launch_bounds__(1024) void ivfpq_tau_kernel(gnnlm_ivfpq_tau_t p) {
    constexpr int CB = 2 * HIST_BINS;                                        // combined bins (list offsets up to HIST_BINS bins)
    __shared__ int comb[CB];
    __shared__ int part[1024];
    __shared__ int bstar_s;
    const int tid = threadIdx.x;
    const int64_t q = blockIdx.x;
    const float delta = p.qmeta[q * 4], sum_lo = p.qmeta[q * 4 + 1], amax = p.qmeta[q * 4 + 2];
    float bmin = INFINITY, bmax = -INFINITY;
    for (int d = 0; d < p.D; ++d) {
        if (p.probe_list[q * p.ld_probe + d] < 0) continue;
        const float b = p.probe_bias[q * p.ld_probe + d];
        bmin = fminf(bmin, b);
        bmax = fmaxf(bmax, b);
    }
    int c0 = 0, c1 = 0;                                                      // combined bins tid and tid + 1024
    for (int d = 0; d < p.D; ++d) {
        if (p.probe_list[q * p.ld_probe + d] < 0) continue;
        // floor of the bias difference in units of delta, a little low on purpose (a smaller offset only lowers the bound)
        const float offf = (p.probe_bias[q * p.ld_probe + d] - bmin) / delta * 0.9999f - 1.f;
        const int sh = (int)fminf(fmaxf(offf, 0.f), (float)(HIST_BINS << HIST_SHIFT)) >> HIST_SHIFT;
        const uint32_t* h = p.hist + (q * p.D + d) * HIST_BINS;
        const int b0 = tid - sh, b1 = tid + 1024 - sh;
        if (b0 >= 0 && b0 < HIST_BINS) c0 += (int)h[b0];
        if (b1 >= 0 && b1 < HIST_BINS) c1 += (int)h[b1];
    }
    comb[tid] = c0;
    comb[tid + 1024] = c1;
    if (tid == 0) bstar_s = -1;
    __syncthreads();
    // b*: the highest combined bin whose suffix count reaches k.  Thread t owns bins [2 t, 2 t + 2); suffix sums over the threads
    // by a doubling scan in LDS, then the one thread whose pair holds the crossing picks the bin
    const int h0 = comb[2 * tid], h1 = comb[2 * tid + 1], mine = h0 + h1;
    part[tid] = mine;
    __syncthreads();
    int suf = mine;                                                          // sum of part[t .. 1023]
    for (int step = 1; step < 1024; step <<= 1) {
        const int other = tid + step < 1024 ? part[tid + step] : 0;
        __syncthreads();
        suf += other;
        part[tid] = suf;
        __syncthreads();
    }
    if (suf >= p.k && suf - mine < p.k) bstar_s = (suf - mine + h1 >= p.k) ? 2 * tid + 1 : 2 * tid;   // exactly one thread
    __syncthreads();
    if (tid == 0) {
        float tau = -INFINITY;
        if (bstar_s >= 0) {
            const float eps = 66.f * 2.3841858e-7f * (64.f * amax + fmaxf(fabsf(bmin), fabsf(bmax)));
            tau = (bmin + sum_lo) + (float)(bstar_s << HIST_SHIFT) * delta * (1.f - 6.1035156e-5f) - eps;   // delta (1 - 2^-14) < 1 / inv
            tau -= fabsf(tau) * 2.3841858e-7f + 1e-30f;                       // strictly below the k keys' scores: candidates are score > tau
        }
        p.tau[q] = tau;
    }
}

// A TIGHTER threshold from the survivors themselves, then the survivors that can still beat it (round 4).  The threshold pass's
// tau is a lower bound of the k-th best score from the first D lists, histogrammed in bins of 16: the filter lets ~6 x k keys
// through and ~4 x k of them score above tau.  Every survivor carries its integer sum (as the excess over its query's integer
// threshold), i.e. a lower bound LB(x) = bias_l + sum_lo + sum_u(x) delta' - eps <= score(x) over ALL probed lists, un-binned:
// the k-th largest LB is a valid and much tighter threshold tau1 (at least k keys score >= it), and only the survivors whose
// UPPER bound exceeds tau1 (the filter's own integer test, with tau1) need an exact score.  One workgroup per query:
// (1) the k-th largest LB to the resolution of a 2048-bin histogram over [tau, max LB] (one pass: a lower edge is a valid threshold too),
// (2) compaction of the records in place, count in `out_cnt` (surv_cnt keeps the filter's count: the overflow check reads it),
// tau[q] = max(tau, tau1).  The search result is unchanged: candidates stay a superset of the true k best.
#ifndef GNNLM_REFINE_NT
#define GNNLM_REFINE_NT 1024     // threads per query (A/B: 512 was slower, 0.46 against 0.40 ms)
#endif
// (a workgroup's refinement as a function: its own kernel below, and the prologue of the fused refine + re-score launch.  `hist`:
// 2048 ints of LDS the caller lends.  Returns the records left and the threshold to every thread.)
template <int EPT>
__device__ __forceinline__ void refine_wg(const gnnlm_ivfpq_refine_t& p, const int64_t q, int* hist, int& n_left, float& tau_left) {
    constexpr int NT = GNNLM_REFINE_NT, NB = 2048, NWV = NT / 64;
    __shared__ int wtot[NWV];
    __shared__ int dig_s, base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = min(p.surv_cnt[q * SURV_CNT_STRIDE], p.cap);
    uint2* surv = reinterpret_cast<uint2*>(p.surv) + q * p.cap;
    const float tau0 = p.tau[q];
    n_left = n;
    tau_left = tau0;
    if (n < p.k || !(tau0 > -INFINITY)) {                                    // fewer survivors than k, or no threshold at all (the sums say nothing)
        if (tid == 0 && p.out_cnt) p.out_cnt[q * SURV_CNT_STRIDE] = n;
        return;
    }
    const float delta = p.qmeta[q * 4], sum_lo = p.qmeta[q * 4 + 1], amax = p.qmeta[q * 4 + 2];
    const float dlo = delta * (1.f - 6.1035156e-5f);                         // delta (1 - 2^-14) < 1 / inv (ivfpq_tau_kernel)
    const float* coarse = p.coarse + q * p.ld_coarse;
    // LB of a record: bias_l + sum_lo + sum_u delta' - eps (eps: the float32 rounding of the score chain), strictly below the score
    auto lower_bound = [&](const uint2& rec) -> float {
        const int list = (int)(rec.y & ((1u << SURV_LIST_BITS) - 1u));
        int su = (int)(rec.y >> SURV_LIST_BITS);
        const float bias = coarse[list];
        if (su >= SURV_SUM_BIG) su = min(16320, max(0, filter_threshold(p.qmeta, q, bias, tau0) + SURV_EXCESS_MAX + 128 * 64));   // (rare: the very best keys)
        const float eps = 66.f * 2.3841858e-7f * (64.f * amax + fabsf(bias));
        const float lb = (bias + sum_lo) + (float)su * dlo - eps;
        return lb - (fabsf(lb) * 2.3841858e-7f + 1e-30f);                    // candidates are score > tau
    };
    // the first EPT * NT records live in registers (one memory round trip); longer lists (a doubled capacity) re-read the rest
    constexpr int C0 = EPT * NT;
    uint2 rrec[EPT];
    float rlb[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) rrec[e] = surv[min(tid + e * NT, n - 1)];
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        rlb[e] = tid + e * NT < n ? lower_bound(rrec[e]) : -INFINITY;
        mx = fmaxf(mx, rlb[e]);
    }
    for (int c = C0 + tid; c < n; c += NT) mx = fmaxf(mx, lower_bound(surv[c]));
    // ---- (1) the k-th largest LB, to the resolution of 2048 bins over [tau0, max LB] (far finer than the 64-delta slack of the
    // bounds; a float radix select of the exact value serialises on LDS atomics: the bounds of a query share their leading bits)
    __shared__ float wmax[NWV];
    mx = wave_max(mx);
    if (lane == 0) wmax[wave] = mx;
    __syncthreads();
    mx = wmax[0];
    for (int w = 1; w < NWV; ++w) mx = fmaxf(mx, wmax[w]);
    const float span = mx - tau0;
    float tau1 = tau0;
    if (span > 0.f) {                                                        // (else no bound above the old threshold: nothing to gain)
        const float inv_w = (float)NB / span * (1.f - 1e-6f);
        for (int e = tid; e < NB; e += NT) hist[e] = 0;
        __syncthreads();
        auto bin_of = [&](float lb) -> int { return lb >= tau0 ? min(NB - 1, (int)((lb - tau0) * inv_w)) : -1; };
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int bn = bin_of(rlb[e]);
            if (bn >= 0) atomicAdd(&hist[bn], 1);
        }
        for (int c = C0 + tid; c < n; c += NT) {
            const int bn = bin_of(lower_bound(surv[c]));
            if (bn >= 0) atomicAdd(&hist[bn], 1);
        }
        __syncthreads();
        // the highest bin whose suffix count reaches k (thread t owns bins 2 t, 2 t + 1; suffix sums by shuffles, then over the waves)
        constexpr int BPT = NB / NT;
        int own[BPT], mine = 0;
#pragma unroll
        for (int bb = 0; bb < BPT; ++bb) { own[bb] = hist[BPT * tid + bb]; mine += own[bb]; }
        int suf = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_down(suf, d, 64);
            if (lane + d < 64) suf += o;
        }
        if (lane == 0) wtot[wave] = suf;
        if (tid == 0) dig_s = -1;
        __syncthreads();
        for (int w = wave + 1; w < NWV; ++w) suf += wtot[w];
        if (suf >= p.k && suf - mine < p.k) {
            int acc_ = suf - mine;
#pragma unroll
            for (int bb = BPT - 1; bb >= 0; --bb) {
                if (acc_ + own[bb] >= p.k) { dig_s = BPT * tid + bb; break; }
                acc_ += own[bb];
            }
        }
        __syncthreads();
        if (dig_s > 0) {
            // every bound of bins >= dig_s is >= tau0 + dig_s / inv_w: at least k keys score above it (two roundings of slack)
            const float t = tau0 + (float)dig_s / inv_w * (1.f - 2e-6f);
            tau1 = fmaxf(tau0, t - fabsf(t) * 4.7683716e-7f);
        }
    }
    // ---- (2) keep the records whose upper bound exceeds tau1.  A record is dropped only if its UPPER bound (bias + sum_lo +
    // (sum_u + 64) delta + eps, the filter's) lies below tau1 with a margin of 0.1 % of the sum term + 2 delta for this formula's
    // own rounding: a slightly weaker test than the filter's integer one
    auto keep_rec = [&](const uint2& rec) -> bool {
        const int list = (int)(rec.y & ((1u << SURV_LIST_BITS) - 1u)), su = (int)(rec.y >> SURV_LIST_BITS);
        if (su >= SURV_SUM_BIG) return true;
        const float bias = coarse[list];
        const float eps = 66.f * 2.3841858e-7f * (64.f * amax + fabsf(bias) + fabsf(tau1));
        const float ub = (bias + sum_lo) + (float)(su + 66) * delta * 1.001f + eps;
        return !(ub < tau1);
    };
    // the register-resident records: every thread counts what it keeps, one exclusive scan over the threads, then the writes
    // (all reads happened long ago: in place is safe)
    int kept = 0;
    uint32_t kmask = 0u;
#pragma unroll
    for (int e = 0; e < EPT; ++e)
        if (tid + e * NT < n && keep_rec(rrec[e])) { kmask |= 1u << e; ++kept; }
    int incl = kept;                                                         // inclusive prefix over the lanes of the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    __syncthreads();                                                         // (wtot is free again)
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int at = incl - kept;
    for (int w = 0; w < wave; ++w) at += wtot[w];
    int total = 0;
    for (int w = 0; w < NWV; ++w) total += wtot[w];
#pragma unroll
    for (int e = 0; e < EPT; ++e)
        if (kmask >> e & 1u) surv[at++] = rrec[e];
    // the rest of a longer list, chunk by chunk behind what has been written (write index <= read index)
    if (n > C0) {
        if (tid == 0) base_s = total;
        __syncthreads();
        for (int c0 = C0; c0 < n; c0 += NT) {
            const int c = c0 + tid;
            uint2 rec = uint2{0u, 0u};
            bool keep = false;
            if (c < n) { rec = surv[c]; keep = keep_rec(rec); }
            const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
            const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            __syncthreads();
            if (lane == 0) wtot[wave] = __builtin_popcountll(m);
            __syncthreads();                                                 // (every read of the chunk is done: the writes stay behind them)
            int before = base_s;
            for (int w = 0; w < wave; ++w) before += wtot[w];
            if (keep) surv[before + rank] = rec;
            __syncthreads();
            if (tid == 0) { int tot = 0; for (int w = 0; w < NWV; ++w) tot += wtot[w]; base_s += tot; }
            __syncthreads();
        }
        total = base_s;
    }
    if (tid == 0) {
        if (p.out_cnt) p.out_cnt[q * SURV_CNT_STRIDE] = total;
        p.tau[q] = tau1;
    }
    n_left = total;
    tau_left = tau1;
}

template <int EPT>
__global__ __launch_bounds__(GNNLM_REFINE_NT) void ivfpq_refine_kernel(gnnlm_ivfpq_refine_t p) {
    __shared__ int hist[2048];
    int n_left;
    float tau_left;
    refine_wg<EPT>(p, (int64_t)blockIdx.x, hist, n_left, tau_left);
}

// Exact float32 scores of the survivors, in the summation order of the f32 scan (ivfpq.hip scan_rot: look-up s of half h
// goes to sub-quantizer 32 h + (row + s) % 32, even look-ups into one chain, odd ones into the other, halves in order,
// score = bias + (chain0 + chain1)).  One workgroup per query, its table in LDS.
#ifndef GNNLM_RESCORE_NT
#define GNNLM_RESCORE_NT 1024   // threads per query (A/B, medians of 7: 1024: 1.14 ms, 512: 1.13, 256 with two workgroups per CU: 1.34)
#endif
// (a workgroup's re-score as a function: `n` records of query q against the threshold `tau`; TAB_READY: the query's table is already
// on its way into rtab by LDS-DMA -- the fused launch below -- and the caller has waited for it)
template <int M, bool TAB_READY>
__device__ __forceinline__ void rescore_wg(const gnnlm_ivfpq_rescore_t& p, const int64_t q, const int n, const float tau, float* rtab) {
    __shared__ int ccnt;
    constexpr int NT = GNNLM_RESCORE_NT;                                    // 16 waves: the survivors' code rows are random 64-B reads
    uint32_t* cdw = reinterpret_cast<uint32_t*>(rtab + M * 256);            // dword k of thread t at [k][t]: bank = t mod 32 for stores and byte reads alike
    const int tid = threadIdx.x;
#ifdef GNNLM_RESCORE_DBG
    // instrumented build (tools/rescore_phases.py): thread 0's clock64 deltas per phase in the last 8 slots of the query's candidate row
    long long dbg_t = clock64(); int dbg_i = 0;
    auto dbg = [&]() { const long long t = clock64(); if (tid == 0) reinterpret_cast<int*>(p.cand_val + (q + 1) * p.cand_cap - 8)[dbg_i] = (int)(t - dbg_t); ++dbg_i; dbg_t = t; };
#else
    auto dbg = [&]() {};
#endif
    if (tid == 0) ccnt = 0;
    const uint2* surv = reinterpret_cast<const uint2*>(p.surv) + q * p.cap;
    // Three loads deep, in an order that never drains the queue (s_waitcnt vmcnt counts in issue order): survivor i's code row and
    // bias are consumed while row i + 1 and the RECORD {row, list} of survivor i + 2 are in flight; a step issues record i + 3 first,
    // then bias and row i + 2 (their addresses come from a record that is one step old: waiting for it leaves the younger loads in
    // flight).  Over a loop's back edge the compiler's wait insertion loses the issue order and drains the queue at the loop head
    // (vmcnt(0)); the first 16 steps (16384 survivors: every capacity the search uses by default) are therefore unrolled -- straight-
    // line code, exact waits, no copies between the register sets.  Indices beyond n are clamped (unconditional loads).
    const int last = max(n - 1, 0);
    auto record = [&](int e, uint2& rec) __attribute__((always_inline)) { rec = surv[min(e, last)]; };
    auto rowload = [&](const uint2& rec, v4u (&cr)[M / 16], float& bias) __attribute__((always_inline)) {
        bias = p.coarse[q * p.ld_coarse + (rec.y & ((1u << SURV_LIST_BITS) - 1u))];   // (first: the youngest loads of a step are the row's)
        const v4u* crow = reinterpret_cast<const v4u*>(p.codes + (int64_t)rec.x * M);
#pragma unroll
        for (int c16 = 0; c16 < M / 16; ++c16) cr[c16] = crow[c16];
    };
    uint2 recC = uint2{0u, 0u}, recD = recC;                                // records in flight (alternating)
    v4u cr0[M / 16], cr1[M / 16];                                           // code rows: two register sets, alternating -- no copies
    float bias0 = 0.f, bias1 = 0.f;
    uint32_t rid0 = 0u, rid1 = 0u;
    if (n > 0) {
        if (!TAB_READY) {
            const float4* src = reinterpret_cast<const float4*>(p.lut + q * p.ld_lut);
            float4* dst = reinterpret_cast<float4*>(rtab);
            for (int e = tid; e < M * 64; e += NT) dst[e] = src[e];
        }
        uint2 r0, r1;
        record(tid, r0);
        record(tid + NT, r1);
        record(tid + 2 * NT, recC);
        rowload(r0, cr0, bias0);
        rid0 = r0.x;
        rowload(r1, cr1, bias1);
        rid1 = r1.x;
    }
    dbg();                                                                   // 0: table copy, first records and rows issued
    __syncthreads();
    dbg();                                                                   // 1: barrier
    const uint8_t* cb = reinterpret_cast<const uint8_t*>(cdw + tid);        // byte m of the row: cb[(m >> 2) * 4 * NT + (m & 3)]
    // one survivor: its row (set `cr`) -> LDS, the next loads issued (record i + 3 into rec_new, then row i + 2 -- named by rec_old,
    // which arrived a step ago -- into the set just freed), then the 64 look-ups
    auto one = [&](int e, v4u (&cr)[M / 16], float& bset, uint32_t& rset, const uint2& rec_old, uint2& rec_new) __attribute__((always_inline)) {
        const int64_t row = rset;
        const float bias = bset;
#pragma unroll
        for (int c16 = 0; c16 < M / 16; ++c16) {
            cdw[(4 * c16 + 0) * NT + tid] = cr[c16].x;
            cdw[(4 * c16 + 1) * NT + tid] = cr[c16].y;
            cdw[(4 * c16 + 2) * NT + tid] = cr[c16].z;
            cdw[(4 * c16 + 3) * NT + tid] = cr[c16].w;
        }
        record(e + 3 * NT, rec_new);
        rowload(rec_old, cr, bset);
        rset = rec_old.x;
        const int rot = (int)(row & 31);
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int h = 0; h < M / 32; ++h) {
#pragma unroll 8
            for (int s = 0; s < 32; s += 2) {
                const int m0 = 32 * h + ((rot + s) & 31), m1 = 32 * h + ((rot + s + 1) & 31);
                a0 = a0 + rtab[m0 * 256 + cb[(m0 >> 2) * (4 * NT) + (m0 & 3)]];
                a1 = a1 + rtab[m1 * 256 + cb[(m1 >> 2) * (4 * NT) + (m1 & 3)]];
            }
        }
        const float score = bias + (a0 + a1);
        if (score > tau) {
            const int pos = atomicAdd(&ccnt, 1);
            if (pos < p.cand_cap) {
                p.cand_val[q * p.cand_cap + pos] = score;
                p.cand_id[q * p.cand_cap + pos] = row;                       // the payload follows below: a load here would make the loop
            }                                                                // wait for it AND for the code rows requested before it
        }
    };
    int e = tid;
#pragma unroll
    for (int it = 0; it < 8; ++it) {                                         // 16 steps, straight-line
        if (e >= n) break;
        one(e, cr0, bias0, rid0, recC, recD);
        e += NT;
        if (e >= n) break;
        one(e, cr1, bias1, rid1, recD, recC);
        e += NT;
    }
    for (; e < n; e += 2 * NT) {                                             // beyond 16384 survivors (a doubled capacity): the rolled loop
        one(e, cr0, bias0, rid0, recC, recD);
        if (e + NT >= n) break;
        one(e + NT, cr1, bias1, rid1, recD, recC);
    }
    dbg();                                                                   // 2: thread 0's records
    __syncthreads();
    dbg();                                                                   // 3: the other waves' records
    if (tid == 0) p.cand_cnt[q] = ccnt;
    // rows -> payloads (ids, labels): every thread's loads independent of each other, nothing else in flight
    const int nc = min(ccnt, p.cand_cap);
    // (loading the payload with every row instead, and the first records without waiting for the count, moved the time into the first
    // phase and added requests: 1.22 against 1.14 ms -- the kernel runs at ~0.7 of the memory system's request rate, ~3 per record)
    for (int e = tid; e < nc; e += NT) p.cand_id[q * p.cand_cap + e] = p.payload[p.cand_id[q * p.cand_cap + e]];
    dbg();                                                                   // 4: payloads
}

template <int M>
__global__ __launch_bounds__(GNNLM_RESCORE_NT) void ivfpq_rescore_kernel(gnnlm_ivfpq_rescore_t p) {
    extern __shared__ __attribute__((aligned(16))) float rtab[];            // [M][256] f32, then the threads' code rows [M / 4 dwords][NT]
    const int64_t q = blockIdx.x;
    rescore_wg<M, false>(p, q, min(p.surv_cnt[q * SURV_CNT_STRIDE], p.cap), p.tau[q], rtab);
}

// Refinement + re-score of a query in ONE launch (round 6; gnnlm_ivfpq_rescore with `qmeta`).  The two kernels were one workgroup
// per query each, both waiting on memory most of their time (the re-score 40 % of its 35 us for its 64-KiB table to arrive: copied
// through registers behind a barrier -- tools/rescore_phases.py).  Here the table is requested FIRST, by LDS-DMA (no registers, nothing
// waits for it), the refinement runs while it arrives (its histogram in the code rows' staging area), and the re-score starts on the
// records the refinement has just compacted (in place, in the query's survivor list: this workgroup's own writes, ordered by the
// barrier).  Same arithmetic, same order: the candidates are those of the two-launch path.
static_assert(GNNLM_REFINE_NT == GNNLM_RESCORE_NT, "the fused refine + re-score launch runs both on one workgroup shape");
template <int EPT>
__global__ __launch_bounds__(GNNLM_RESCORE_NT) void ivfpq_refine_rescore_kernel(gnnlm_ivfpq_rescore_t p) {
    constexpr int M = 64;
    extern __shared__ __attribute__((aligned(16))) float rtab[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t q = blockIdx.x;
    const int n0 = min(p.surv_cnt[q * SURV_CNT_STRIDE], p.cap);
    if (n0 > 0) {
        // [64][256] f32 = 64 pieces of 1 KiB: wave w brings pieces 4 w .. 4 w + 3 (global_load_lds_dwordx4: 16 B per lane, lane order;
        // M0 = the piece's LDS byte address).  From inline asm: the compiler's counters do not see them -- its own waits only ever
        // get more conservative by that -- and would otherwise guard every LDS read of the refinement with vmcnt(0)
        const float* src = p.lut + q * p.ld_lut + (4 * wave) * 256 + 4 * lane;
        const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)rtab + (unsigned)(4 * __builtin_amdgcn_readfirstlane(wave)) * 1024u;
        unsigned keep_;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep_) : "v"(src), "s"(dst) : "memory");
    }
    gnnlm_ivfpq_refine_t r;
    r.surv = const_cast<uint32_t*>(p.surv);  r.surv_cnt = p.surv_cnt;  r.out_cnt = p.out_cnt;  r.cap = p.cap;
    r.tau = const_cast<float*>(p.tau);  r.qmeta = p.qmeta;  r.coarse = p.coarse;  r.ld_coarse = p.ld_coarse;  r.n = p.n;  r.k = p.k;
    int n_left;
    float tau_left;
    refine_wg<EPT>(r, q, reinterpret_cast<int*>(rtab + M * 256), n_left, tau_left);   // (histogram: where the code rows are staged afterwards)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // this wave's table pieces have landed (and its compacted records have left)
    __syncthreads();                                                         // ... everybody's
    rescore_wg<M, true>(p, q, n_left, tau_left, rtab);
}

}  // namespace

int ivfpq_pack_tiles(const uint8_t* codes, int64_t N, int M, uint8_t* out, hipStream_t stream) {
    GNNLM_REQUIRE(codes && out && N >= 0 && M == 64, "ivfpq_pack_tiles: need M = 64");
    GNNLM_REQUIRE((uintptr_t)out % 16 == 0, "ivfpq_pack_tiles: 16-byte aligned output");
    const int64_t threads = ((N + 15) >> 4 << 4) * 4;
    if (threads == 0) return OK;
    GNNLM_REQUIRE(cdiv(threads, (int64_t)256) < (1ll << 31), "ivfpq_pack_tiles: too many rows for one launch");
    hipLaunchKernelGGL(pack_tiles_kernel, dim3((unsigned)cdiv(threads, (int64_t)256)), dim3(256), 0, stream, codes, N, out);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int ivfpq_build_groups(const int64_t* pl, int64_t ld, int64_t n, int P, int nlist, int64_t seg, int32_t* grp_list, int32_t* grp_q,
                       int64_t* grp_out, int32_t* n_groups, int32_t* scratch, hipStream_t stream) {
    GNNLM_REQUIRE(n >= 0 && P > 0 && nlist > 0 && n * P < (1ll << 31) && ld >= P, "ivfpq_build_groups: bad shape");
    GNNLM_REQUIRE(pl && grp_list && grp_q && n_groups && scratch, "ivfpq_build_groups: null operand");
    const int64_t G = n * P / QG + nlist + 1, pairs = n * P;
    const int64_t fill = std::max<int64_t>(G * QG, 2 * (int64_t)(nlist + 1));
    hipLaunchKernelGGL(groups_fill_kernel, dim3((unsigned)cdiv(fill, (int64_t)256)), dim3(256), 0, stream, grp_list, grp_q, grp_out, G, scratch, nlist);
    if (pairs > 0)
        hipLaunchKernelGGL(groups_count_kernel, dim3((unsigned)cdiv(pairs, (int64_t)256)), dim3(256), 0, stream, pl, ld, n, P, nlist, scratch);
    hipLaunchKernelGGL(groups_scan_kernel, dim3(1), dim3(1024), 0, stream, scratch, nlist, n_groups);
    if (pairs > 0)
        hipLaunchKernelGGL(groups_scatter_kernel, dim3((unsigned)cdiv(pairs, (int64_t)256)), dim3(256), 0, stream, pl, ld, n, P, nlist, seg, scratch,
                           scratch + nlist + 1, grp_list, grp_q, grp_out);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int ivfpq_quantize_lut(const float* lut, int64_t ld_lut, int64_t n, int M, uint8_t* qlut, float* qmeta, hipStream_t stream) {
    GNNLM_REQUIRE(lut && qlut && qmeta && n >= 0 && n < (1ll << 31) && M == 64 && ld_lut >= 64 * 256 && ld_lut % 4 == 0 &&
                      (uintptr_t)lut % 16 == 0 && (uintptr_t)qlut % 16 == 0,
                  "ivfpq_quantize_lut: need M = 64 and 16-byte aligned tables");
    if (n == 0) return OK;
    hipLaunchKernelGGL(quantize_lut_kernel, dim3((unsigned)n), dim3(256), 0, stream, lut, ld_lut, qlut, qmeta);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int ivfpq_scan8(const gnnlm_ivfpq_scan8_t& d, hipStream_t stream) {
    GNNLM_REQUIRE(d.max_groups >= 0, "ivfpq_scan8: bad group count");
    if (d.max_groups == 0) return OK;
    GNNLM_REQUIRE(d.tiles && d.list_off && d.qlut && d.qmeta && d.coarse && d.grp_list && d.grp_q && d.n_groups &&
                      (d.out_hist || (d.tau && d.surv && d.surv_cnt && d.cap > 0)),
                  "ivfpq_scan8: null operand");
    GNNLM_REQUIRE(d.M == 64 && (uintptr_t)d.tiles % 16 == 0 && (uintptr_t)d.qlut % 4 == 0, "ivfpq_scan8: need M = 64, aligned images");
    GNNLM_REQUIRE(!d.out_hist || d.grp_out, "ivfpq_scan8: out_hist needs grp_out");
    GNNLM_REQUIRE(d.nlist > 0 && d.nlist <= (1 << SURV_LIST_BITS) && d.max_list >= 0 && d.max_list < (1ll << SURV_ROW_BITS),
                  "ivfpq_scan8: the survivor records hold 18 bits of list and 19 bits of row: need 0 < nlist <= 2^18 and max_list < 2^19");
    GNNLM_LDS_OPT_IN(&ivfpq_scan8_kernel<false>, SCAN_LDS);
    GNNLM_LDS_OPT_IN(&ivfpq_scan8_kernel<true>, SUMS_LDS);
    ProfScope prof(d.out_hist ? K_IVF8S : K_IVF8, stream, 0.0, 0.0);   // work figures are device-side (list lengths): bench.py computes them
    int dev = 0, cus = 0;
    GNNLM_HIP(hipGetDevice(&dev));
    GNNLM_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int64_t grid = std::min<int64_t>(8 * cdiv((int64_t)d.max_groups, (int64_t)8), 8 * cdiv((int64_t)std::max(cus, 8), (int64_t)8));   // persistent: one per CU
    if (d.out_hist) hipLaunchKernelGGL(ivfpq_scan8_kernel<true>, dim3((unsigned)grid), dim3(NTH), SUMS_LDS, stream, d);
    else hipLaunchKernelGGL(ivfpq_scan8_kernel<false>, dim3((unsigned)grid), dim3(NTH), SCAN_LDS, stream, d);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int ivfpq_tau(const gnnlm_ivfpq_tau_t& d, hipStream_t stream) {
    GNNLM_REQUIRE(d.n >= 0 && d.n < (1ll << 31) && d.D >= 1 && d.k > 0, "ivfpq_tau: bad shape");
    if (d.n == 0) return OK;
    GNNLM_REQUIRE(d.hist && d.probe_list && d.probe_bias && d.qmeta && d.tau && d.ld_probe >= d.D, "ivfpq_tau: null operand");
    ProfScope prof(K_TAU, stream, 0.0, 4.0 * (double)d.n * d.D * HIST_BINS);
    hipLaunchKernelGGL(ivfpq_tau_kernel, dim3((unsigned)d.n), dim3(1024), 0, stream, d);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

// the search's results: payload (id << label_bits | label, -1 = none) -> id in place, label to out_vals (none: `val_last`, what
// numpy's vals[-1] reads for the reference, knn/knn_model.py:198)
__global__ __launch_bounds__(256) void split_payload_kernel(int64_t* __restrict__ idx, int64_t n, int label_bits, int32_t val_last, int32_t* __restrict__ out_vals) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int64_t v = idx[e];
    idx[e] = v < 0 ? v : v >> label_bits;
    if (out_vals) out_vals[e] = v < 0 ? val_last : (int32_t)(v & ((1ll << label_bits) - 1));
}
int ivfpq_split_payload(int64_t* idx, int64_t n, int label_bits, int32_t val_last, int32_t* out_vals, hipStream_t stream) {
    GNNLM_REQUIRE(idx && n >= 0 && label_bits > 0 && label_bits < 32, "ivfpq_split_payload: bad arguments");
    if (n == 0) return OK;
    hipLaunchKernelGGL(split_payload_kernel, dim3((unsigned)cdiv(n, (int64_t)256)), dim3(256), 0, stream, idx, n, label_bits, val_last, out_vals);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int ivfpq_refine(const gnnlm_ivfpq_refine_t& d, hipStream_t stream) {
    GNNLM_REQUIRE(d.n >= 0 && d.n < (1ll << 31) && d.k > 0 && d.cap > 0, "ivfpq_refine: bad shape");
    if (d.n == 0) return OK;
    GNNLM_REQUIRE(d.surv && d.surv_cnt && d.out_cnt && d.tau && d.qmeta && d.coarse, "ivfpq_refine: null operand");
    ProfScope prof(K_TAU, stream, 0.0, 16.0 * (double)d.n * d.k);
    // 16 records per thread in registers (the default capacity of 16384 records; what is beyond is re-read)
    hipLaunchKernelGGL(ivfpq_refine_kernel<16>, dim3((unsigned)d.n), dim3(GNNLM_REFINE_NT), 0, stream, d);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int ivfpq_rescore(const gnnlm_ivfpq_rescore_t& d, hipStream_t stream) {
    GNNLM_REQUIRE(d.n >= 0 && d.n < (1ll << 31), "ivfpq_rescore: bad query count");
    if (d.n == 0) return OK;
    GNNLM_REQUIRE(d.codes && d.payload && d.lut && d.coarse && d.tau && d.surv && d.surv_cnt && d.cand_val && d.cand_id && d.cand_cnt &&
                      d.cap > 0 && d.cand_cap > 0,
                  "ivfpq_rescore: null operand");
    GNNLM_REQUIRE((d.M == 64 || d.M == 32) && d.ld_lut >= (int64_t)d.M * 256 && d.ld_lut % 4 == 0 && (uintptr_t)d.lut % 16 == 0 &&
                      (uintptr_t)d.codes % 16 == 0,
                  "ivfpq_rescore: need M = 32 or 64, 16-byte aligned tables");
    ProfScope prof(K_RESCORE, stream, 0.0, 0.0);
    const size_t lds = (size_t)d.M * 256 * 4 + GNNLM_RESCORE_NT * (size_t)d.M;
    if (d.qmeta) {                                                          // refinement + re-score in one launch (ABI 10)
        GNNLM_REQUIRE(d.M == 64 && d.k > 0, "ivfpq_rescore: the fused refinement needs M = 64 and k > 0");
        GNNLM_LDS_OPT_IN(&ivfpq_refine_rescore_kernel<8>, lds);
        hipLaunchKernelGGL(ivfpq_refine_rescore_kernel<8>, dim3((unsigned)d.n), dim3(GNNLM_RESCORE_NT), lds, stream, d);
        GNNLM_LAUNCH_CHECK();
        return OK;
    }
    if (d.M == 64) {
        GNNLM_LDS_OPT_IN(&ivfpq_rescore_kernel<64>, lds);
        hipLaunchKernelGGL(ivfpq_rescore_kernel<64>, dim3((unsigned)d.n), dim3(GNNLM_RESCORE_NT), lds, stream, d);
    } else {
        GNNLM_LDS_OPT_IN(&ivfpq_rescore_kernel<32>, lds);
        hipLaunchKernelGGL(ivfpq_rescore_kernel<32>, dim3((unsigned)d.n), dim3(GNNLM_RESCORE_NT), lds, stream, d);
    }
    GNNLM_LAUNCH_CHECK();
    return OK;
}

}  // namespace gnnlm
