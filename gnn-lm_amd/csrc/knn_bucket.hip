// kNN interpolation with ids-only search results, BUCKETED (round 4): the label gather `vals[knns]` of knn/knn_model.py:198 without
// paying one memory request per look-up.
//
// The one-pass kernel (rowops.hip, knn_interp_regs_kernel) issues k random reads per token into a label table of hundreds of MB;
// every one of them misses the L2 and costs one request to the fabric, and the memory system serves ~48-51 G requests/s whatever
// their size (tools/probes/fetch_calib.hip): 8192 x 1024 look-ups = 176 us, at that ceiling.  Here the look-ups are ROUTED to where
// their table bytes are:
//   pass A  (one workgroup per 16 tokens) reads the ids once, packs every look-up as {row, tag of the token's target, j, token} and
//           sorts the tile by REGION of the one-byte tag table (gnnlm_label_tags; row >> shift, <= 1024 regions of <= 128 KB) in LDS --
//           one histogram, ranks from returning LDS atomics -- and appends each region's run to that region's list in HBM with
//           coalesced stores (one global atomic per region and tile); it also writes the `id != -1` bits the softmax needs;
//   pass B  (one workgroup per region) copies the region's slice of the tag table into LDS with coalesced loads -- the table is read
//           ONCE, as a stream -- and looks the region's entries up THERE; a tag equal to the target's (1 in 256, plus the true hits)
//           sets candidate bit j of the token;
//   pass C  (one wave per token) confirms the token's few candidates with the 4-byte label and is otherwise the one-pass kernel's
//           arithmetic with the hit bit in place of `vals[id] == target`: the same operations in the same order, hence the same bits.
// Lists have a fixed capacity (1.5 x the uniform share + slack); a run that does not fit is looked up by pass A on the spot (one
// request each, like the one-pass kernel).  First formulation (regions of 1 MB looked up through the L2 of one XCD by persistent
// workgroups): 60-95 us for pass B alone -- the L1 takes ~4 cycles per missing line however near the data is.
#include "kernels.h"

namespace gnnlm {
namespace {

#ifndef GNNLM_KB_EXP
#define GNNLM_KB_EXP 0                    // timing-only ablations (wrong results): 1 pass A without its global stores, 2 without its global atomics,
#endif                                    //   4 pass B without the confirmations, 8 without the slice copy, 16 without the entry loads
constexpr int MAX_REGIONS = 1024;         // regions of the label table
constexpr int MAX_SHIFT = 17;             // a region's slice of the tag table: at most 128 KB (it lives in LDS in pass B)
#ifndef GNNLM_KB_TILE
#define GNNLM_KB_TILE 16                  // (A/B: 8 -> 139 us per call against 112, 4 -> 246: a tile's run per region must fill a line)
#endif
constexpr int TILE_TOK = GNNLM_KB_TILE, A_NT = 64 * TILE_TOK;   // pass A: tokens per workgroup (one wave each)
constexpr int B_NT = 1024, B_EPT = 16;    // pass B: threads per region, entries a thread keeps in flight

struct BucketPlan {
    int shift, n_regions;                 // region of a row = row >> shift
    int64_t cap;                          // entries per region list
    int32_t* cursor;                      // [MAX_REGIONS] entries appended to each list
    uint32_t* hit;                        // [n, 32] bit j of token i: the tag of vals[ids[i, j]] equals the tag of targets[i] (a candidate)
    uint32_t* valid;                      // [n, 32] bit j: ids[i, j] != -1
    uint64_t* entries;                    // [n_regions, cap] {row : 30 | target's tag : 8 | j : 10 | token : 16}
};

__device__ __forceinline__ uint32_t label_tag32(int64_t v) { return ((uint32_t)v * 2654435761u) >> 24; }
// entry = {row : 30 | tag of the token's target : 8 | j : 10 | token : 16}: the one scattered read of a look-up is the row's tag byte
__device__ __forceinline__ uint64_t pack_entry(int64_t row, uint32_t ttag, int j, int64_t tok) {
    return (uint64_t)row << 34 | (uint64_t)ttag << 26 | (uint64_t)j << 16 | (uint64_t)tok;
}
__device__ __forceinline__ int64_t entry_row(uint64_t e) { return (int64_t)(e >> 34); }

// a look-up whose tag equals the target's (1 in 256 + the true hits) becomes a CANDIDATE bit of its token: pass C, which has the
// token's target at hand, reads the 4-byte label of its few candidates (confirming here put two dependent random reads at the end of
// every workgroup of pass B: 22 us of the pass)
__device__ __forceinline__ void confirm_entry(const KnnInterpParams& p, const BucketPlan& b, uint64_t e) {
    const int j = (int)(e >> 16) & 1023;
    const int64_t tok = (int64_t)(e & 0xffffu);
    atomicOr(&b.hit[tok * 32 + (j >> 5)], 1u << (j & 31));
}

__global__ __launch_bounds__(A_NT) void knn_bucket_pass_a(KnnInterpParams p, BucketPlan b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* sorted = reinterpret_cast<uint64_t*>(smem);                        // [TILE_TOK * 1024] the tile's entries, sorted by region
    int* hist = reinterpret_cast<int*>(smem + (size_t)TILE_TOK * 1024 * 8);      // [MAX_REGIONS] entries of the tile per region
    int* bstart = hist + MAX_REGIONS;                                            // [MAX_REGIONS + 1] first position of a region in `sorted`
    int* gbase = bstart + MAX_REGIONS + 1;                                       // [MAX_REGIONS] first position of the tile's run in the region's global list
    __shared__ int wsum[A_NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t tok = (int64_t)blockIdx.x * TILE_TOK + wave;
    for (int e = tid; e < MAX_REGIONS; e += A_NT) hist[e] = 0;
    // the wave's token: look-up j = lane + 64 t lives in (lane, t), as in pass C
    int64_t row[16];
    int rank[16];
    const bool live = tok < p.n;
    const int64_t* ids = p.ids + (live ? tok : 0) * p.k;
    const uint32_t ttag = label_tag32(p.targets[live ? tok : 0]);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int j = lane + 64 * t;
        const int64_t id = (live && j < p.k) ? ids[j] : -1;
        const uint64_t m = __builtin_amdgcn_ballot_w64(live && j < p.k && id != -1);
        if (live && lane == 0) *reinterpret_cast<uint64_t*>(b.valid + tok * 32 + 2 * t) = m;
        // numpy indexing semantics of vals[knns]: -1 wraps to the last row; rows outside the shard / the store match nothing
        const int64_t r = (id < 0 ? id + p.n_store : id) - p.row0;
        row[t] = (live && j < p.k && r >= 0 && r < p.n_local) ? r : -1;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; ++t) rank[t] = row[t] >= 0 ? atomicAdd(&hist[(int)(row[t] >> b.shift)], 1) : 0;
    __syncthreads();
    {   // exclusive scan of the MAX_REGIONS counts: BPT consecutive ones per thread, shuffles inside a wave, then over the waves
        constexpr int BPT = MAX_REGIONS / A_NT;
        static_assert(BPT * A_NT == MAX_REGIONS, "whole region counts per thread");
        int own[BPT], mine = 0;
#pragma unroll
        for (int x = 0; x < BPT; ++x) { own[x] = hist[BPT * tid + x]; mine += own[x]; }
        int incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int at = incl - mine;
        for (int w = 0; w < wave; ++w) at += wsum[w];
#pragma unroll
        for (int x = 0; x < BPT; ++x) {
            bstart[BPT * tid + x] = at;
            at += own[x];
            // (the cursor keeps counting past the capacity: pass B clamps)
            gbase[BPT * tid + x] = (GNNLM_KB_EXP & 2) ? (int)blockIdx.x * 20 : (own[x] > 0 ? atomicAdd(&b.cursor[BPT * tid + x], own[x]) : 0);
        }
        if (tid == A_NT - 1) bstart[MAX_REGIONS] = at;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; ++t)
        if (row[t] >= 0) sorted[bstart[(int)(row[t] >> b.shift)] + rank[t]] = pack_entry(row[t], ttag, lane + 64 * t, tok);
    __syncthreads();
    // copy-out in sorted order: consecutive lanes, consecutive positions of a region's list (a run is a dozen entries: one line).
    // (Tried: 4-byte entries in LDS -- 76 KB, two workgroups per CU -- with a region-by-region copy-out: 68 us against 49, a run
    // fills a quarter of a wave's store.)
    const int total = bstart[MAX_REGIONS];
    for (int s = tid; s < total; s += A_NT) {
        const uint64_t e = sorted[s];
        const int bin = (int)(entry_row(e) >> b.shift);
        const int64_t pos = (int64_t)gbase[bin] + (s - bstart[bin]);
        if (GNNLM_KB_EXP & 1) { if (e == 12345ull) b.entries[0] = e; }
        else if (pos < b.cap) b.entries[(int64_t)bin * b.cap + pos] = e;         // (every position below min(cursor, cap) gets written)
        else if ((uint32_t)p.vals_tag[entry_row(e)] == ((uint32_t)(e >> 26) & 255u)) confirm_entry(p, b, e);   // no room in the list: looked up here and now
    }
}

// one workgroup per region: its slice of the tag table -> LDS (coalesced), then the region's entries against it
__global__ __launch_bounds__(B_NT) void knn_bucket_pass_b(KnnInterpParams p, BucketPlan b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char slice[];
    const int r = blockIdx.x, tid = threadIdx.x;
    const int64_t cnt = (GNNLM_KB_EXP & 3) ? min((int64_t)10600, b.cap) : min((int64_t)b.cursor[r], b.cap);
    if (cnt == 0) return;
    const int64_t row_lo = (int64_t)r << b.shift;
    const int64_t bytes = min((int64_t)1 << b.shift, p.n_local - row_lo);
    const uint64_t* list = b.entries + (int64_t)r * b.cap;
    // the first entries of every thread are requested before the slice: their latency hides behind its copy
    uint64_t e[B_EPT];
#pragma unroll
    for (int x = 0; x < B_EPT; ++x) {
        const int64_t c = tid + (int64_t)x * B_NT;
        e[x] = c < cnt ? ((GNNLM_KB_EXP & 16) ? (uint64_t)(row_lo + (c * 7919 & 0xffff)) << 34 : list[c]) : ~0ull;
    }
    const uint8_t* src = p.vals_tag + row_lo;
    if (GNNLM_KB_EXP & 8) { if (tid == 0) slice[0] = src[0]; }
    else if (((uintptr_t)src & 15) == 0) {                                       // (always, unless the table is tiny: row_lo = r << shift)
        const int64_t n16 = bytes >> 4;
        const uint4* s16 = reinterpret_cast<const uint4*>(src);
        for (int64_t o = tid; o < n16; o += B_NT) reinterpret_cast<uint4*>(slice)[o] = s16[o];
        for (int64_t o = (n16 << 4) + tid; o < bytes; o += B_NT) slice[o] = src[o];
    } else {
        for (int64_t o = tid; o < bytes; o += B_NT) slice[o] = src[o];
    }
    __syncthreads();
#pragma unroll
    for (int x = 0; x < B_EPT; ++x)
        if (e[x] != ~0ull && (uint32_t)slice[entry_row(e[x]) - row_lo] == ((uint32_t)(e[x] >> 26) & 255u) + ((GNNLM_KB_EXP & 4) ? 999u : 0u)) confirm_entry(p, b, e[x]);
    for (int64_t c = tid + (int64_t)B_EPT * B_NT; c < cnt; c += B_NT) {          // a list longer than 16 k entries
        const uint64_t ee = list[c];
        if ((uint32_t)slice[entry_row(ee) - row_lo] == ((uint32_t)(ee >> 26) & 255u)) confirm_entry(p, b, ee);
    }
}

// the arithmetic of knn_interp_regs_kernel<16> (rowops.hip), operation by operation, with the masks of passes A and B
__global__ __launch_bounds__(256) void knn_bucket_pass_c(KnnInterpParams p, BucketPlan b, float log_1ml, float log_l) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= p.n) return;
    const float* sims = p.sims + i * p.k;
    float sv[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int j = lane + 64 * t;
        sv[t] = j < p.k ? sims[j] : 0.f;
    }
    uint32_t vw[16], hw[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        vw[t] = b.valid[i * 32 + 2 * t + (lane >> 5)];
        hw[t] = b.hit[i * 32 + 2 * t + (lane >> 5)];
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const bool is_id = (vw[t] >> (lane & 31)) & 1u;                          // ids[j] != -1
        sv[t] = (is_id ? sv[t] : -1e10f) / p.temperature;
        if (lane + 64 * t < p.k) mx = fmaxf(mx, sv[t]);
    }
    mx = wave_max(mx);
    // the token's candidates (tag matches: ~k / 256 + the true hits): the label decides
    const int64_t tgt = p.targets[i];
    uint32_t hitm = 0u;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        if ((hw[t] >> (lane & 31)) & 1u) {
            const int64_t id = p.ids[i * p.k + lane + 64 * t];
            const int64_t row = (id < 0 ? id + p.n_store : id) - p.row0;
            const int64_t lab = p.vals_itemsize == 2 ? (int64_t) reinterpret_cast<const int16_t*>(p.vals)[row]
                                                     : (int64_t) reinterpret_cast<const int32_t*>(p.vals)[row];
            if (lab == tgt) hitm |= 1u << t;
        }
    }
    float den = 0.f, num = 0.f;
    int rec = 0;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        if (lane + 64 * t < p.k) {
            const float e = expf(sv[t] - mx);
            const bool hit = (hitm >> t) & 1u;
            den += e;
            num += hit ? e : 0.f;
            rec += hit;
        }
    }
    den = wave_sum(den);
    num = wave_sum(num);
    rec = (int)wave_sum((float)rec);
    if (lane == 0) {
        const float pk = num / den;
        if (p.out_pknn) p.out_pknn[i] = pk;
        if (p.out_recall) p.out_recall[i] = rec;
        const float a = p.lm_logp[i] + log_1ml;
        const float bb = logf(pk + 1e-10f) + log_l;
        const float m = fmaxf(a, bb);
        p.out_logp[i] = m + logf(expf(a - m) + expf(bb - m));
    }
}

int plan_shift(int64_t n_local) {
    int s = 0;
    while (((n_local - 1) >> s) >= MAX_REGIONS) ++s;
    return s;
}
int64_t plan_cap(int64_t n, int k, int n_regions) { return ((n * k / n_regions) * 3 / 2 + 4096 + 511) / 512 * 512; }
int plan_regions(int64_t n_local) { return (int)(((n_local - 1) >> plan_shift(n_local)) + 1); }

}  // namespace

// scratch layout: cursors | hit masks | valid masks | (256-byte aligned) the region lists
static size_t entries_offset(int64_t n) { return ((size_t)MAX_REGIONS * 4 + (size_t)n * 32 * 4 * 2 + 255) / 256 * 256; }

size_t knn_interp_scratch_bytes(int64_t n, int k, int64_t n_local) {
    if (n <= 0 || k <= 0 || n_local <= 0) return 0;
    const int nr = plan_regions(n_local);
    return entries_offset(n) + (size_t)nr * plan_cap(n, k, nr) * 8;
}

bool knn_interp_bucketed_eligible(const KnnInterpParams& p) {
    return !p.knn_vals && p.vals && p.vals_tag && p.scratch && p.k <= 1024 && p.n <= (1ll << 16) && p.n_local > 0 &&
           plan_shift(p.n_local) <= MAX_SHIFT && p.scratch_bytes >= knn_interp_scratch_bytes(p.n, p.k, p.n_local) && (uintptr_t)p.scratch % 16 == 0;
}

int knn_interp_bucketed(const KnnInterpParams& p, float log_1ml, float log_l, hipStream_t stream) {
    BucketPlan b;
    b.shift = plan_shift(p.n_local);
    b.n_regions = plan_regions(p.n_local);
    b.cap = plan_cap(p.n, p.k, b.n_regions);
    unsigned char* s = static_cast<unsigned char*>(p.scratch);
    b.cursor = reinterpret_cast<int32_t*>(s);
    b.hit = reinterpret_cast<uint32_t*>(s + MAX_REGIONS * 4);
    b.valid = b.hit + p.n * 32;
    b.entries = reinterpret_cast<uint64_t*>(s + entries_offset(p.n));
    GNNLM_HIP(hipMemsetAsync(s, 0, (size_t)MAX_REGIONS * 4 + (size_t)p.n * 32 * 4, stream));      // cursors, hit masks
    const size_t lds_a = (size_t)TILE_TOK * 1024 * 8 + (size_t)(MAX_REGIONS + MAX_REGIONS + 1 + MAX_REGIONS) * 4;
    GNNLM_LDS_OPT_IN(&knn_bucket_pass_a, lds_a);
    hipLaunchKernelGGL(knn_bucket_pass_a, dim3((unsigned)cdiv(p.n, (int64_t)TILE_TOK)), dim3(A_NT), lds_a, stream, p, b);
    const size_t lds_b = ((size_t)1 << b.shift) + 32;
    GNNLM_LDS_OPT_IN(&knn_bucket_pass_b, ((size_t)1 << MAX_SHIFT) + 32);
    hipLaunchKernelGGL(knn_bucket_pass_b, dim3((unsigned)b.n_regions), dim3(B_NT), lds_b, stream, p, b);
    hipLaunchKernelGGL(knn_bucket_pass_c, dim3((unsigned)cdiv(p.n, (int64_t)4)), dim3(256), 0, stream, p, b, log_1ml, log_l);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

}  // namespace gnnlm
