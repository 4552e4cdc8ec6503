// Group assignment on the device (ABI 9, include/gnnlm.h: gnnlm_group_assign).
//
// The ntgt pipeline of a multi-layer HGT runs once per DISTINCT centre row of a batch (the reference's own
// "todo: merge same nodes", fairseq/data/token_block_dataset.py:355; a group's states depend on its centre row only because
// ntgt nodes never receive from tgt nodes, :395-398) and, with the cross-batch cache, only for the rows the cache lacks.
// Deciding which rows those are used to be `torch.unique` (a radix sort) plus one host round trip for the count; here it is a
// handful of HBM-bound passes over the batch's ids against a direct row -> slot table (int32 per datastore row), claimed with
// atomicCAS, and the count never leaves the device: every consumer kernel reads it (gnnlm_hgt_io_t.n_unique_dev).
//
// New rows are appended in ARRIVAL order (one wave-aggregated atomicAdd per wave).  The order is not deterministic -- and does
// not matter: every kernel downstream computes a row from that row's inputs alone (fixed k order inside the GEMMs), so the
// states are bit-identical wherever a group lands.  tests/test_hgt_gpu.py asserts it.
#include "kernels.h"

namespace gnnlm {
namespace {

enum { CS_GEN = 0, CS_FILL0 = 1, CS_FILL1 = 2, CS_SWITCHES = 3, CS_COMPUTED = 4 };     // cache_state[]
enum { CT_N = 0, CT_FLIP = 1, CT_DROP = 2 };                                          // counters[]
constexpr int32_t FREE = -1, CLAIMED = -2;

__device__ __forceinline__ int64_t half_lo(const gnnlm_group_assign_t& p, int gen) { return gen ? p.cache_cap / 2 : 0; }
__device__ __forceinline__ int64_t half_cap(const gnnlm_group_assign_t& p, int gen) { return gen ? p.cache_cap - p.cache_cap / 2 : p.cache_cap / 2; }

__global__ __launch_bounds__(256) void ga_init_kernel(gnnlm_group_assign_t p) {
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    for (int64_t i = i0; i < p.n; i += stride) p.group_ids[i] = -1;
    if (i0 < 4) p.counters[i0] = 0;
}

// Claim every row of the batch that has no slot yet: the first thread to turn its table entry from FREE into CLAIMED appends the
// row to group_ids.  `only_if_flip`: the second pass after a generation switch (rows whose slots were just dropped).
__global__ __launch_bounds__(256) void ga_claim_kernel(gnnlm_group_assign_t p, int only_if_flip) {
    if (only_if_flip && p.counters[CT_FLIP] == 0) return;
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * 256;
    // (uniform trip count per wave: the ballot below needs every lane)
    for (int64_t base = (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63); base < p.n; base += stride) {
        const int64_t i = base + lane;
        const int64_t row = i < p.n ? p.ids[i] : -1;
        bool mine = false;
        if (row >= 0 && row < p.n_store && p.slot_of[row] == FREE)
            mine = atomicCAS(&p.slot_of[row], FREE, CLAIMED) == FREE;
        const unsigned long long m = __ballot(mine);
        if (m) {
            int first = 0;
            if (lane == 0) first = atomicAdd(&p.counters[CT_N], __popcll(m));
            first = __shfl(first, 0);
            if (mine) p.group_ids[first + __popcll(m & ((1ull << lane) - 1))] = row;
        }
    }
}

// One thread: does the half being filled have room for the batch's new rows?  If not, the other half is emptied (ga_drop) and
// becomes the one being filled.
__global__ void ga_decide_kernel(gnnlm_group_assign_t p) {
    int32_t* cs = p.cache_state;
    int gen = cs[CS_GEN] & 1;
    if ((int64_t)p.counters[CT_N] > half_cap(p, gen) - cs[CS_FILL0 + gen]) {
        gen ^= 1;
        cs[CS_GEN] = gen;
        p.counters[CT_FLIP] = 1;
        p.counters[CT_DROP] = cs[CS_FILL0 + gen];
        cs[CS_FILL0 + gen] = 0;
        cs[CS_SWITCHES] += 1;
    }
}
__global__ __launch_bounds__(256) void ga_drop_kernel(gnnlm_group_assign_t p) {
    if (p.counters[CT_FLIP] == 0) return;
    const int64_t lo = half_lo(p, p.cache_state[CS_GEN] & 1), n = p.counters[CT_DROP];
    for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < n; s += (int64_t)gridDim.x * 256) p.slot_of[p.id_of_slot[lo + s]] = FREE;
}
// Slots of the new rows: cache mode -> the next free slots of the half being filled; merge mode -> the position in group_ids.
__global__ __launch_bounds__(256) void ga_assign_kernel(gnnlm_group_assign_t p) {
    const int64_t n = p.counters[CT_N];
    int64_t lo = 0;
    if (p.cache_cap > 0) {
        const int gen = p.cache_state[CS_GEN] & 1;
        lo = half_lo(p, gen) + p.cache_state[CS_FILL0 + gen];
    }
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < n; g += (int64_t)gridDim.x * 256) {
        const int64_t row = p.group_ids[g];
        const int32_t slot = (int32_t)(lo + g);
        p.slot_of[row] = slot;
        if (p.cache_cap > 0) {
            p.id_of_slot[slot] = row;
            p.group_slot[g] = slot;
        }
    }
}
__global__ __launch_bounds__(256) void ga_index_kernel(gnnlm_group_assign_t p) {
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (int64_t i = i0; i < p.n; i += (int64_t)gridDim.x * 256) {
        const int64_t row = p.ids[i];
        p.group_index[i] = (row >= 0 && row < p.n_store) ? p.slot_of[row] : -1;
    }
    if (i0 == 0 && p.cache_cap > 0) {          // (ga_assign, which reads the fill, has finished: stream order)
        int32_t* cs = p.cache_state;
        cs[CS_FILL0 + (cs[CS_GEN] & 1)] += p.counters[CT_N];
        cs[CS_COMPUTED] = (int32_t)(((int64_t)cs[CS_COMPUTED] + p.counters[CT_N]) & 0x7fffffff);
    }
}
// merge mode: hand the table back clean
__global__ __launch_bounds__(256) void ga_reset_kernel(gnnlm_group_assign_t p) {
    const int64_t n = p.counters[CT_N];
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < n; g += (int64_t)gridDim.x * 256) p.slot_of[p.group_ids[g]] = FREE;
}

// rows of window w (groups [w * per, (w + 1) * per) of *n_dev groups) times mult = 1 .. 8: counts[w * 8 + mult - 1]
__global__ void window_counts_kernel(const int32_t* n_dev, int64_t cap, int64_t per, int n_win, int32_t* counts) {
    const int t = threadIdx.x;
    if (t >= n_win * 8) return;
    const int w = t / 8, mult = t % 8 + 1;
    const int64_t n = min((int64_t)*n_dev, cap);
    const int64_t in_w = max((int64_t)0, min(per, n - (int64_t)w * per));
    counts[t] = (int32_t)(in_w * mult);
}
// code row of neighbour e in a fetched buffer: the centre slot of its group
__global__ __launch_bounds__(256) void nb_code_rows_kernel(const int32_t* group_index, const int32_t* fetched_index, int n_g, int64_t n, int32_t* out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int64_t g = group_index[e];
    out[e] = g < 0 ? -1 : (fetched_index ? fetched_index[g * n_g] : (int32_t)(g * n_g));
}
// dst[slots[g]] = src[(index ? index[g * stride] : g * stride)] for g < *n_dev: rows of `bytes` bytes (a multiple of 16)
__global__ __launch_bounds__(256) void scatter_code_rows_kernel(const uint8_t* src, const int32_t* index, int64_t stride, const uint8_t* valid,
                                                                uint8_t* dst, const int32_t* slots, const int32_t* n_dev, int64_t cap, int bytes) {
    const int per_row = bytes >> 4;
    const int64_t n = (n_dev ? min((int64_t)*n_dev, cap) : cap) * per_row;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int64_t g = e / per_row;
        const int part = (int)(e - g * per_row);
        const int64_t slot = slots[g];
        if (slot < 0) continue;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (!valid || valid[g * stride]) {
            const int64_t r = index ? (int64_t)index[g * stride] : g * stride;
            v = *reinterpret_cast<const uint4*>(src + r * bytes + 16 * part);
        }
        *reinterpret_cast<uint4*>(dst + slot * bytes + 16 * part) = v;
    }
}

// ABI 11 (row-keyed K / V of layer 0).  Slot s = g * n_g + c of the first *n_dev groups -> its datastore row, by gather_decode's own
// rule (centre first, then o - left .. o - 1, then o + 1 .. o + right; -1: no such row, or a group beyond the count)
__global__ __launch_bounds__(256) void slot_rows_kernel(const int64_t* centres, int64_t G, const int32_t* n_dev, int left, int right, int reach, int64_t n_store, int64_t* out) {
    const int n_g = 1 + left + right;
    const int64_t live = n_dev ? min(G, (int64_t)*n_dev) : G;
    for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < G * n_g; s += (int64_t)gridDim.x * 256) {
        const int64_t g = s / n_g;
        const int c = (int)(s - g * n_g);
        int64_t row = -1;
        if (g < live) {
            const int64_t centre = centres[g];
            const int delta = c == 0 ? 0 : (c <= left ? c - 1 - left : c - left);
            row = centre + delta;
            if (!(centre >= 0 && row >= 0 && row < n_store) || delta > reach || -delta > reach) row = -1;      // (`reach`: only the slots whose K / V a later layer reads)
        }
        out[s] = row;
    }
}
}  // namespace

int slot_rows(const int64_t* centres, int64_t G, const int32_t* n_dev, int left, int right, int reach, int64_t n_store, int64_t* out, hipStream_t stream) {
    if (G == 0) return OK;
    hipLaunchKernelGGL(slot_rows_kernel, dim3((unsigned)std::min<int64_t>(cdiv(G * (1 + left + right), (int64_t)256), 4096)), dim3(256), 0, stream,
                       centres, G, n_dev, left, right, reach, n_store, out);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int group_assign(const gnnlm_group_assign_t& p, hipStream_t stream) {
    GNNLM_REQUIRE(p.n >= 0 && p.n < (1ll << 31) && p.n_store > 0 && p.slot_of && p.counters, "group_assign: bad arguments");
    GNNLM_REQUIRE(p.n == 0 || (p.ids && p.group_ids && p.group_index), "group_assign: null ids / outputs");
    const bool cache = p.cache_cap > 0;
    GNNLM_REQUIRE(!cache || (p.id_of_slot && p.cache_state && p.group_slot && p.cache_cap < (1ll << 31) && p.n <= p.cache_cap / 2),
                  "group_assign: cache mode needs id_of_slot, cache_state, group_slot and n <= cache_cap / 2");
    ProfScope prof(K_MISC, stream, 0.0, 28.0 * p.n);
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(p.n, 256), 2048));
    hipLaunchKernelGGL(ga_init_kernel, dim3(blocks), dim3(256), 0, stream, p);
    if (p.n > 0) {
        hipLaunchKernelGGL(ga_claim_kernel, dim3(blocks), dim3(256), 0, stream, p, 0);
        if (cache) {
            hipLaunchKernelGGL(ga_decide_kernel, dim3(1), dim3(1), 0, stream, p);
            hipLaunchKernelGGL(ga_drop_kernel, dim3((unsigned)std::min<int64_t>(cdiv(p.cache_cap, 512), 2048)), dim3(256), 0, stream, p);
            hipLaunchKernelGGL(ga_claim_kernel, dim3(blocks), dim3(256), 0, stream, p, 1);
        }
        hipLaunchKernelGGL(ga_assign_kernel, dim3(blocks), dim3(256), 0, stream, p);
        hipLaunchKernelGGL(ga_index_kernel, dim3(blocks), dim3(256), 0, stream, p);
        if (!cache) hipLaunchKernelGGL(ga_reset_kernel, dim3(blocks), dim3(256), 0, stream, p);
    }
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int window_counts(const int32_t* n_dev, int64_t cap, int64_t per, int n_win, int32_t* counts, hipStream_t stream) {
    GNNLM_REQUIRE(n_dev && counts && n_win >= 1 && n_win <= 128 && per > 0, "window_counts: bad arguments");
    hipLaunchKernelGGL(window_counts_kernel, dim3(1), dim3(1024), 0, stream, n_dev, cap, per, n_win, counts);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int nb_code_rows(const int32_t* group_index, const int32_t* fetched_index, int n_g, int64_t n, int32_t* out, hipStream_t stream) {
    if (n == 0) return OK;
    hipLaunchKernelGGL(nb_code_rows_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream, group_index, fetched_index, n_g, n, out);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int scatter_code_rows(const uint8_t* src, const int32_t* index, int64_t stride, const uint8_t* valid, uint8_t* dst, const int32_t* slots,
                      const int32_t* n_dev, int64_t cap, int bytes, hipStream_t stream) {
    GNNLM_REQUIRE(src && dst && slots && bytes > 0 && bytes % 16 == 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0,
                  "scatter_code_rows: rows must be multiples of 16 bytes, 16-byte aligned");
    if (cap == 0) return OK;
    hipLaunchKernelGGL(scatter_code_rows_kernel, dim3((unsigned)std::min<int64_t>(cdiv(cap * (bytes >> 4), 256), 2048)), dim3(256), 0, stream,
                       src, index, stride, valid, dst, slots, n_dev, cap, bytes);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

}  // namespace gnnlm
