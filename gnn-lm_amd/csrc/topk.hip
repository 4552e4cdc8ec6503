// Running top-k over score chunks: the selection half of the on-device kNN search
// (replaces faiss `index.search`'s k-selection, knn/knn_model.py:87-101 and knn/find_knn.py:55-70; the scores come
// from the f32 MFMA GEMM over a chunk of keys, or from the IVF-PQ scan).
//
// One workgroup per query.  The running state (best_val / best_id, k entries, best first) lives in HBM between
// calls; a call folds one chunk of `ncols` scores into it.  Only scores that beat the current k-th best survive the
// scan (after the first chunks that is a handful per row), are appended to an LDS candidate buffer, sorted with a
// bitonic network and merged into the sorted state with one bitonic merge.  No [n, N] score matrix, no global sort.
// Order: better value first; equal values by ascending id (what a stable argsort of the full row gives), so the
// result does not depend on the chunking.
#include <cmath>
#include <cstdlib>

#include "kernels.h"

#ifndef GNNLM_TOPK_PRESEL
#define GNNLM_TOPK_PRESEL 2      // counting pre-pass from this many times KP columns on (first chunk)
#endif
namespace gnnlm {
namespace {

// true if a ranks before b
__device__ __forceinline__ bool before(float av, int64_t ai, float bv, int64_t bi) { return av > bv || (av == bv && ai < bi); }

// compare-exchange network helpers over LDS arrays val[], id[]; `n` threads cooperate, L is a power of two
template <int NT>
__device__ __forceinline__ void bitonic_merge_desc(float* val, int64_t* id, int L, int first_stride, int tid) {
    for (int stride = first_stride; stride > 0; stride >>= 1) {
        for (int e = tid; e < L / 2; e += NT) {
            const int i = ((e & ~(stride - 1)) << 1) | (e & (stride - 1));
            const int j = i + stride;
            const float a = val[i], b = val[j];
            const int64_t ia = id[i], ib = id[j];
            if (before(b, ib, a, ia)) { val[i] = b; val[j] = a; id[i] = ib; id[j] = ia; }
        }
        __syncthreads();
    }
}
// full bitonic sort, best first
template <int NT>
__device__ __forceinline__ void bitonic_sort_desc(float* val, int64_t* id, int L, int tid) {
    for (int size = 2; size <= L; size <<= 1) {
        // first step of each stage pairs i with its mirror inside the block of `size` (makes every block sorted best-first)
        for (int e = tid; e < L / 2; e += NT) {
            const int blk = e / (size / 2), off = e % (size / 2);
            const int i = blk * size + off, j = blk * size + size - 1 - off;
            const float a = val[i], b = val[j];
            const int64_t ia = id[i], ib = id[j];
            if (before(b, ib, a, ia)) { val[i] = b; val[j] = a; id[i] = ib; id[j] = ia; }
        }
        __syncthreads();
        bitonic_merge_desc<NT>(val, id, L, size / 4, tid);
    }
}

struct TopkParams {
    const float* scores; int64_t ld; int64_t n; int ncols;
    int64_t col0; const int64_t* col_ids; const float* col_scale; const float* col_bias; float alpha;
    int k, largest, init;
    float* best_val; int64_t* best_id;
    const int32_t* row_ncols;       // optional [n]: row r only has its first row_ncols[r] columns (ragged candidate lists)
    const int64_t* ids; int64_t ld_ids;
};

// KP: power of two >= k.  LDS: val[2 KP] + id[2 KP]; [0, KP) the state (padded with -inf), [KP, 2 KP) the candidates.
template <int KP>
__global__ __launch_bounds__(256) void topk_merge_kernel(TopkParams p) {
    constexpr int NT = 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int64_t* id = reinterpret_cast<int64_t*>(smem_raw);                 // [2 KP]
    float* val = reinterpret_cast<float*>(id + 2 * KP);                 // [2 KP]
    __shared__ int cnt;
    const int tid = threadIdx.x;
    const int64_t row = blockIdx.x;
    const float sign = p.largest ? 1.f : -1.f;
    const float NEG = -INFINITY;

    for (int e = tid; e < KP; e += NT) {
        const bool have = !p.init && e < p.k;
        val[e] = have ? sign * p.best_val[row * p.k + e] : NEG;
        id[e] = have ? p.best_id[row * p.k + e] : (int64_t)-1;
        if (have && id[e] < 0) val[e] = NEG;                            // unfilled slots of an earlier call
    }
    if (tid == 0) cnt = 0;
    __syncthreads();
    float tau = p.k <= KP ? val[p.k - 1] : NEG;                         // current k-th best (-inf while fewer than k)

    auto flush = [&]() {
        // candidates [KP, KP + cnt) -> sort best-first, reverse into a bitonic sequence with the state, merge
        const int m = cnt;
        for (int e = m + tid; e < KP; e += NT) { val[KP + e] = NEG; id[KP + e] = (int64_t)0x7fffffffffffffffll; }
        __syncthreads();
        bitonic_sort_desc<NT>(val + KP, id + KP, KP, tid);
        // reverse the candidate half in place -> [state best-first | candidates worst-first] is bitonic
        for (int e = tid; e < KP / 2; e += NT) {
            const int i = KP + e, j = 2 * KP - 1 - e;
            const float a = val[i]; val[i] = val[j]; val[j] = a;
            const int64_t ia = id[i]; id[i] = id[j]; id[j] = ia;
        }
        __syncthreads();
        bitonic_merge_desc<NT>(val, id, 2 * KP, KP, tid);
        if (tid == 0) cnt = 0;
        __syncthreads();
        tau = val[p.k - 1];
    };

    const float* srow = p.scores + row * p.ld;
    const int ncols = p.row_ncols ? min(p.ncols, p.row_ncols[row]) : p.ncols;
    // Wide first chunks (the dense round of an IVF search: k = 1024 of ~50 k scores) would pass ~k (1 + ln(n / k)) candidates
    // through ~8 sort-and-merge rounds before the running threshold tightens.  A counting pre-pass bounds the threshold
    // first: a histogram of the row over NB value bins (range from a sample, out-of-range values clamp: the bin function is
    // monotone whatever the sample says), b* = the highest bin with at least k values in bins >= b*; the selection pass then
    // only admits values of bins >= b* -- every one of the k best is among them, ~k plus one bin's population in all.
    constexpr int NB = 1024;
    __shared__ int hist[NB];
    __shared__ int rng[2];                           // order-preserving integer images of the sample's min / max
    __shared__ int bstar_s;
    const bool presel = p.init && ncols >= GNNLM_TOPK_PRESEL * KP;
    float b_lo = 0.f, b_scale = 0.f;
    int bstar = 0;
    auto value_of = [&](int c, float& v, int64_t& cid) {
        v = srow[c] * p.alpha;
        if (p.col_scale) v *= p.col_scale[c];
        if (p.col_bias) v += p.col_bias[c];
        v *= sign;
        cid = p.ids ? p.ids[row * p.ld_ids + c] : (p.col_ids ? p.col_ids[c] : p.col0 + c);
    };
    auto bin_of = [&](float v) { return (int)fminf(fmaxf((v - b_lo) * b_scale, 0.f), (float)(NB - 1)); };
    if (presel) {
        float lo = INFINITY, hi = -INFINITY;
        for (int c = tid; c < min(ncols, 4096); c += NT) {
            float v; int64_t cid;
            value_of(c, v, cid);
            if (cid >= 0 && v > NEG && v < INFINITY) { lo = fminf(lo, v); hi = fmaxf(hi, v); }
        }
        for (int e = tid; e < NB; e += NT) hist[e] = 0;
        if (tid == 0) { rng[0] = 0x7fffffff; rng[1] = (int)0x80000000; }
        __syncthreads();
        // order-preserving integer image of a float: atomicMin / atomicMax on it
        auto ord = [](float x) { const int i = __float_as_int(x); return i >= 0 ? i : i ^ 0x7fffffff; };
        if (lo <= hi) {
            atomicMin(&rng[0], ord(lo));
            atomicMax(&rng[1], ord(hi));
        }
        __syncthreads();
        auto unord = [](int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); };
        const bool have_range = rng[0] <= rng[1];
        const float r_lo = have_range ? unord(rng[0]) : 0.f, r_hi = have_range ? unord(rng[1]) : 0.f;
        b_lo = r_lo;
        b_scale = (r_hi > r_lo) ? (float)NB / (r_hi - r_lo) : 0.f;
        for (int c = tid; c < ncols; c += NT) {
            float v; int64_t cid;
            value_of(c, v, cid);
            if (cid >= 0 && v > NEG) atomicAdd(&hist[bin_of(v)], 1);
        }
        __syncthreads();
        // b*: the highest bin whose suffix count reaches k (0 if the row has fewer than k values): thread t owns bins
        // [4 t, 4 t + 4), suffix sums of the per-thread totals by a scan over LDS
        if (tid == 0) {
            int acc_ = 0, b = NB - 1;
            for (; b > 0; --b) { acc_ += hist[b]; if (acc_ >= p.k) break; }
            bstar_s = b;
        }
        __syncthreads();
        bstar = bstar_s;
    }
    static_assert(KP >= NT, "a sub-block of NT columns must fit the candidate half");
    constexpr int SB = KP / 2 >= NT ? KP / 2 : NT;                      // columns per sub-block: at most SB new candidates
    for (int c0 = 0; c0 < ncols; c0 += SB) {
        // every thread takes its copy of cnt BEFORE any thread of this iteration can bump it (the barrier below separates the
        // read from the atomicAdds): a wave that runs late would otherwise see a larger count, take flush() alone and pair its
        // barriers with the wrong ones
        const int cur = cnt;
        __syncthreads();
        if (cur + SB > KP) flush();
        for (int c = c0 + tid; c < min(ncols, c0 + SB); c += NT) {
            float v;
            int64_t cid;
            value_of(c, v, cid);
            // strictly better than the k-th best, or tied with it and earlier (tau's id is not tracked: keep ties, the merge decides)
            if (cid >= 0 && v >= tau && v > NEG && (!presel || bin_of(v) >= bstar)) {
                const int pos = atomicAdd(&cnt, 1);
                val[KP + pos] = v;
                id[KP + pos] = cid;
            }
        }
        __syncthreads();
    }
    if (cnt > 0) flush();
    for (int e = tid; e < p.k; e += NT) {
        const bool real = val[e] > NEG;
        p.best_val[row * p.k + e] = real ? sign * val[e] : (p.largest ? NEG : INFINITY);
        p.best_id[row * p.k + e] = real ? id[e] : (int64_t)-1;         // faiss pads with -1
    }
}

// ---- the whole row in ONE chunk (init = 1: the final k-selection of the IVF-PQ search over a few thousand candidates per query,
// the probe selection over the coarse scores).  No running state to merge with, so select first, sort once: a radix select on the
// order-preserving integer image of the score (three digit passes: 11 + 11 + 10 bits, histogram in LDS, the digit whose suffix
// count reaches what is still needed) gives the k-th best value; values tied with it are taken by ascending id (a radix select
// over the id's digits, only when ties straddle the cut); the selected entries -- exactly min(k, valid) of them -- go to LDS
// and through one bitonic sort.  Same order as the merge kernel: better value first, equal values by ascending id.
template <int KP, int EPT>
__global__ __launch_bounds__(256) void topk_select_kernel(TopkParams p) {
    constexpr int NT = 256, NB = 2048;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int64_t* id = reinterpret_cast<int64_t*>(smem_raw);                 // [KP]
    float* val = reinterpret_cast<float*>(id + KP);                     // [KP]
    __shared__ int hist[NB];
    __shared__ int wtot[4];
    __shared__ int dig_s, above_s, cnt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row = blockIdx.x;
    const float sign = p.largest ? 1.f : -1.f;
    const float NEG = -INFINITY;
    const float* srow = p.scores + row * p.ld;
    const int ncols = p.row_ncols ? min(p.ncols, p.row_ncols[row]) : p.ncols;
    // column c -> (valid, value, order-preserving key, id); -0 and +0 share a key (they compare equal)
    auto entry = [&](int c, float& v, uint32_t& key, int64_t& cid) -> bool {
        v = srow[c] * p.alpha;
        if (p.col_scale) v *= p.col_scale[c];
        if (p.col_bias) v += p.col_bias[c];
        v = v * sign + 0.f;
        cid = p.ids ? p.ids[row * p.ld_ids + c] : (p.col_ids ? p.col_ids[c] : p.col0 + c);
        const uint32_t u = __float_as_uint(v);
        key = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        return cid >= 0 && v > NEG;
    };
    // the highest digit whose suffix count over hist[0 .. nb) reaches `need` (-1: the whole histogram holds fewer); above_s = the
    // count of the digits above it.  Thread t owns the 8 bins [8 t, 8 t + 8); suffix sums over lanes by shuffles, over waves in LDS
    auto pick = [&](int need) -> int {
        int own[8], mine = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) { own[b] = hist[8 * tid + b]; mine += own[b]; }
        int suf = mine;                                                  // sum over lanes >= lane of this wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_down(suf, d, 64);
            if (lane + d < 64) suf += o;
        }
        if (lane == 0) wtot[wave] = suf;
        if (tid == 0) { dig_s = -1; above_s = 0; }
        __syncthreads();
        for (int w = wave + 1; w < 4; ++w) suf += wtot[w];               // threads >= tid
        if (suf >= need && suf - mine < need) {                          // the crossing is inside this thread's bins: exactly one thread
            int acc_ = suf - mine;
#pragma unroll
            for (int b = 7; b >= 0; --b) {
                if (acc_ + own[b] >= need) { dig_s = 8 * tid + b; above_s = acc_; break; }
                acc_ += own[b];
            }
        }
        __syncthreads();
        return dig_s;
    };
    // The first EPT * 256 columns of the row live in REGISTERS (key and id; every load issued before the first use: one memory
    // round trip instead of one per pass and step -- the passes are latency-bound otherwise); longer rows take the rest from
    // memory in every pass.  key 0 = no entry (valid keys are > key(-inf)).
    uint32_t rkey[EPT];
    int64_t rid[EPT];
    {
        float rv[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int c = tid + e * NT;
            rv[e] = NEG;
            rid[e] = -1;
            if (c < ncols) {
                rv[e] = srow[c];
                rid[e] = p.ids ? p.ids[row * p.ld_ids + c] : (p.col_ids ? p.col_ids[c] : p.col0 + c);
            }
        }
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int c = tid + e * NT;
            float x = rv[e] * p.alpha;
            if (c < ncols) {
                if (p.col_scale) x *= p.col_scale[c];
                if (p.col_bias) x += p.col_bias[c];
            }
            x = x * sign + 0.f;
            const uint32_t u = __float_as_uint(x);
            rkey[e] = (c < ncols && rid[e] >= 0 && x > NEG) ? ((u & 0x80000000u) ? ~u : (u | 0x80000000u)) : 0u;
        }
    }
    auto value_of_key = [](uint32_t key) { return __uint_as_float((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key); };
    constexpr int C0 = EPT * NT;                                         // first column that is not in registers
    uint32_t prefix = 0u, mask = 0u;
    int need = p.k;
    bool all = false;                                                    // fewer valid entries than k: everything is selected
    constexpr int SHIFT[3] = {21, 10, 0}, BITS[3] = {11, 11, 10};
#pragma unroll
    for (int ps = 0; ps < 3; ++ps) {
        for (int e = tid; e < NB; e += NT) hist[e] = 0;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EPT; ++e)
            if (rkey[e] != 0u && (rkey[e] & mask) == prefix) atomicAdd(&hist[(rkey[e] >> SHIFT[ps]) & ((1u << BITS[ps]) - 1u)], 1);
        for (int c = C0 + tid; c < ncols; c += NT) {
            float v; uint32_t key; int64_t cid;
            if (entry(c, v, key, cid) && (key & mask) == prefix) atomicAdd(&hist[(key >> SHIFT[ps]) & ((1u << BITS[ps]) - 1u)], 1);
        }
        __syncthreads();
        const int d = pick(need);
        if (d < 0) { all = true; break; }                                // (only the first pass can come up short)
        need -= above_s;
        prefix |= (uint32_t)d << SHIFT[ps];
        mask |= ((1u << BITS[ps]) - 1u) << SHIFT[ps];
        __syncthreads();
    }
    // `need` of the entries with key == prefix are still to be taken, by ascending id.  hist[d] of the last pass = how many there are
    int64_t id_cut = 0x7fffffffffffffffll;
    if (!all) {
        const int ties = hist[prefix & ((1u << BITS[2]) - 1u)];
        __syncthreads();
        if (ties > need) {                                               // ties straddle the cut: the need-th smallest id among them
            uint64_t ipre = 0ull, imask = 0ull;
            int ineed = need;
            for (int sh = 55; sh >= 0; sh -= 11) {                       // ids are < 2^63: digits of 11 bits from bit 65 down (the top one short)
                const int s2 = sh < 0 ? 0 : sh;
                for (int e = tid; e < NB; e += NT) hist[e] = 0;
                __syncthreads();
#pragma unroll
                for (int e = 0; e < EPT; ++e)
                    if (rkey[e] == prefix && ((uint64_t)rid[e] & imask) == ipre)               // (prefix != 0: a real entry)
                        atomicAdd(&hist[NB - 1 - (int)(((uint64_t)rid[e] >> s2) & (NB - 1))], 1);   // reversed digit: smaller ids are "better"
                for (int c = C0 + tid; c < ncols; c += NT) {
                    float v; uint32_t key; int64_t cid;
                    if (entry(c, v, key, cid) && key == prefix && ((uint64_t)cid & imask) == ipre)
                        atomicAdd(&hist[NB - 1 - (int)(((uint64_t)cid >> s2) & (NB - 1))], 1);
                }
                __syncthreads();
                const int d = pick(ineed);
                ineed -= above_s;
                ipre |= (uint64_t)(NB - 1 - d) << s2;
                imask |= (uint64_t)(NB - 1) << s2;
                __syncthreads();
            }
            id_cut = (int64_t)ipre;                                      // ids are unique per row: exactly `need` tied entries have id <= id_cut
        }
    }
    if (tid == 0) cnt = 0;
    for (int e = tid; e < KP; e += NT) { val[e] = NEG; id[e] = (int64_t)0x7fffffffffffffffll; }
    __syncthreads();
    // two phases: the entries strictly above the cut first, the tied ones (by id) behind them -- with ids that are NOT unique
    // within a row more than `need` tied entries can pass `cid <= id_cut`, and the `pos < KP` guard must then drop tied entries
    // only, never a strictly better one
#pragma unroll 1
    for (int phase = 0; phase < 2; ++phase) {
        if (phase == 1) {
            if (all) break;
            __syncthreads();
        }
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            if (rkey[e] == 0u) continue;
            const bool take = phase == 0 ? (all || rkey[e] > prefix) : (rkey[e] == prefix && rid[e] <= id_cut);
            if (take) {
                const int pos = atomicAdd(&cnt, 1);
                if (pos < KP) { val[pos] = value_of_key(rkey[e]); id[pos] = rid[e]; }
            }
        }
        for (int c = C0 + tid; c < ncols; c += NT) {
            float v; uint32_t key; int64_t cid;
            if (!entry(c, v, key, cid)) continue;
            const bool take = phase == 0 ? (all || key > prefix) : (key == prefix && cid <= id_cut);
            if (take) {
                const int pos = atomicAdd(&cnt, 1);
                if (pos < KP) { val[pos] = v; id[pos] = cid; }
            }
        }
    }
    __syncthreads();
    bitonic_sort_desc<NT>(val, id, KP, tid);
    for (int e = tid; e < p.k; e += NT) {
        const bool real = val[e] > NEG;
        p.best_val[row * p.k + e] = real ? sign * val[e] : (p.largest ? NEG : INFINITY);
        p.best_id[row * p.k + e] = real ? id[e] : (int64_t)-1;
    }
}

}  // namespace

int topk_merge(const gnnlm_topk_t& d, hipStream_t stream) {
    GNNLM_REQUIRE(d.best_val && d.best_id && d.k > 0 && d.k <= 2048, "topk_merge: need state buffers and 0 < k <= 2048");
    GNNLM_REQUIRE(d.n >= 0 && d.ncols >= 0 && d.n < (1ll << 31), "topk_merge: bad shape");
    if (d.n == 0) return OK;
    GNNLM_REQUIRE(d.ncols == 0 || d.scores, "topk_merge: null scores");
    TopkParams p{d.scores, d.ld, d.n, d.ncols, d.col0, d.col_ids, d.col_scale, d.col_bias, d.alpha == 0.f ? 1.f : d.alpha,
                 d.k, d.largest, d.init, d.best_val, d.best_id, d.row_ncols, d.ids, d.ld_ids};
    ProfScope prof(K_TOPK, stream, 0.0, 4.0 * (double)d.n * d.ncols + 24.0 * (double)d.n * d.k);
    const dim3 grid((unsigned)d.n), block(256);
    static const bool merge_only = getenv("GNNLM_TOPK_MERGE_ONLY") != nullptr;       // A/B switch (tests): the merge kernel for everything
    if (d.init && !merge_only && d.ncols <= 16384) {                                  // the whole (not too wide) row at once: select, then one sort;
                                                                                      // wide first chunks (exact search: 65536 columns) keep the counting pre-pass of the merge kernel: 2 passes over the row, not 4
#define GNNLM_TOPK_SELECT(KP)                                                                                          \
    {                                                                                                                  \
        if (d.ncols <= 4096) hipLaunchKernelGGL((topk_select_kernel<KP, 16>), grid, block, (size_t)KP * 12, stream, p); \
        else hipLaunchKernelGGL((topk_select_kernel<KP, 20>), grid, block, (size_t)KP * 12, stream, p);                \
    }
        if (d.k <= 64) GNNLM_TOPK_SELECT(64)
        else if (d.k <= 256) GNNLM_TOPK_SELECT(256)
        else if (d.k <= 1024) GNNLM_TOPK_SELECT(1024)
        else GNNLM_TOPK_SELECT(2048)
#undef GNNLM_TOPK_SELECT
        GNNLM_LAUNCH_CHECK();
        return OK;
    }
#define GNNLM_TOPK_LAUNCH(KP)                                                                                    \
    {                                                                                                            \
        const size_t lds = (size_t)2 * KP * 12;                                                                  \
        hipLaunchKernelGGL((topk_merge_kernel<KP>), grid, block, lds, stream, p);                                \
    }
    if (d.k <= 256) GNNLM_TOPK_LAUNCH(512)          // twice the 256 columns a sub-block may append: small k folds several sub-blocks per sort (probe selection, k = 32 of 4096: 110 -> ~40 us per 1024 queries)
    else if (d.k <= 1024) GNNLM_TOPK_LAUNCH(1024)
    else GNNLM_TOPK_LAUNCH(2048)
#undef GNNLM_TOPK_LAUNCH
    GNNLM_LAUNCH_CHECK();
    return OK;
}

}  // namespace gnnlm
