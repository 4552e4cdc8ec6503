// PQ datastore row gather + decode, HBM-resident.
//
// Replaces, per token block, the host-side row gathers of the reference
//   quant_neighbor_feats[offset] / neighbor_tokens[offset]   fairseq/data/token_block_dataset.py:370-371,391-393
// (a PlasmaArray / np.memmap per-row Python append + np.stack, :407-410) and the table lookup half of
//   TorchPQCodec.decode                                       knn/pq_wrapper.py:169-196
// The neighbour-context expansion of new_build_graph (:378-394) is done in-kernel from the centre row:
// slot c of a group is row  o (c = 0),  o-left .. o-1 (c = 1..left),  o+1 .. o+right (c = left+1..).
//
// HBM-bound: one wave per slot reads the 128-B code row (one cache line) and writes D*4 bytes of
// decoded features, 16 B per lane per store instruction (fully coalesced 1-KiB wave stores).  The
// 1-MiB centroid table is read through L2 (it is shared by every wave on the chip and stays resident).
#include "kernels.h"

namespace gnnlm {
namespace {

__global__ __launch_bounds__(256) void gather_decode_kernel(GatherParams p) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int n_g = 1 + p.left + p.right;
    const int64_t n_slots = (p.n_groups_dev ? min(p.n_groups, (int64_t)*p.n_groups_dev) : p.n_groups) * n_g;      // (device-side group count, ABI 9)
    const int D = p.M * p.dsub;
    const int nq = D / 4;                       // float4 per decoded row
    const int q_per_m = p.dsub / 4;

    for (int64_t s = wave; s < n_slots; s += nwaves) {
        const int64_t g = s / n_g;
        const int c = (int)(s - g * n_g);
        bool local;
        int64_t lrow;
        const uint8_t* srow = nullptr;           // with mapped shards: the row in its owner's memory
        if (p.direct) {
            lrow = p.in_index ? (int64_t)p.in_index[s] : s;
            local = p.in_valid[s] != 0;
        } else {
            const int64_t centre = p.ids[g];
            const int delta = c == 0 ? 0 : (c <= p.left ? c - 1 - p.left : c - p.left);
            const int64_t row = centre + delta;
            const bool valid = centre >= 0 && row >= 0 && row < p.n_store;
            lrow = row - p.row0;
            if (p.shards) {
                srow = valid ? shard_row_ptr(p.shards, row, p.M) : nullptr;
                local = srow != nullptr;
            } else {
                local = valid && lrow >= 0 && lrow < p.n_local;   // sharded store: caller routes ids
            }
        }
        if (lane == 0) {
            if (p.out_valid) p.out_valid[s] = local ? 1 : 0;
            if (p.out_labels) {
                int32_t lab = -1;
                if (local && p.vals)
                    lab = p.vals_itemsize == 2 ? (int32_t) reinterpret_cast<const int16_t*>(p.vals)[lrow]
                                               : reinterpret_cast<const int32_t*>(p.vals)[lrow];
                p.out_labels[s] = lab;
            }
        }
        const uint8_t* crow = srow ? srow : p.codes + (local ? lrow : 0) * p.M;        // (not dereferenced unless local)
        if (p.out_codes) {
            for (int m = lane; m < p.M; m += 64) p.out_codes[s * p.M + m] = local ? crow[m] : 0;
        }
        if (p.out_x) {
            float4* out = reinterpret_cast<float4*>(p.out_x + s * p.ld_x);
            for (int q = lane; q < nq; q += 64) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (local) {
                    const int m = q / q_per_m;
                    const int within = (q - m * q_per_m) * 4;
                    const int code = crow[m];
                    v = *reinterpret_cast<const float4*>(p.centroids + ((int64_t)(m * 256 + code)) * p.dsub + within);
                }
                out[q] = v;
            }
        }
    }
}

// codes / labels only (the owner-side lookup of the sharded store): 8 lanes move one 128-B row with
// 16-B pieces, 8 rows per wave -- the byte-wise path above would spend a whole wave per row.
__global__ __launch_bounds__(256) void gather_rows_kernel(GatherParams p) {
    const int per_row = p.M >> 4;                                   // 16-B pieces per row
    const int64_t n = p.n_groups_dev ? min(p.n_groups, (int64_t)*p.n_groups_dev) : p.n_groups;
    const int64_t total = n * per_row;
    const int64_t stride = (int64_t)gridDim.x * 256;
    // four independent (id -> row piece) chains per thread per trip: the kernel is bound by the latency of the
    // dependent pair of loads, not by bytes
    for (int64_t e0 = (int64_t)blockIdx.x * 256 + threadIdx.x; e0 < total; e0 += 4 * stride) {
        int64_t s[4], lrow[4];
        int part[4];
        bool local[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t e = e0 + u * stride;
            s[u] = e < total ? e / per_row : -1;
            part[u] = (int)(e - s[u] * per_row);
            const int64_t row = s[u] >= 0 ? p.ids[s[u]] : -1;
            lrow[u] = row - p.row0;
            local[u] = s[u] >= 0 && row >= 0 && row < p.n_store && lrow[u] >= 0 && lrow[u] < p.n_local;
        }
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[u] = make_uint4(0, 0, 0, 0);
            if (local[u]) v[u] = *reinterpret_cast<const uint4*>(p.codes + lrow[u] * p.M + 16 * part[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (s[u] < 0) continue;
            *reinterpret_cast<uint4*>(p.out_codes + s[u] * p.M + 16 * part[u]) = v[u];
            if (part[u] == 0) {
                if (p.out_valid) p.out_valid[s[u]] = local[u] ? 1 : 0;
                if (p.out_labels) {
                    int32_t lab = -1;
                    if (local[u] && p.vals)
                        lab = p.vals_itemsize == 2 ? (int32_t) reinterpret_cast<const int16_t*>(p.vals)[lrow[u]]
                                                   : reinterpret_cast<const int32_t*>(p.vals)[lrow[u]];
                    p.out_labels[s[u]] = lab;
                }
            }
        }
    }
}

// PQ encode (TorchPQCodec.encode, knn/pq_wrapper.py:131-167; an offline producer in the reference --
// knn/quantize_features.py:122-146 -- and a "next" row here): codes[r][m] = argmin_c ||c||^2 - 2 x_m . c.
// One wave per row; per sub-quantizer every lane scores 4 of the 256 centroids (32-B rows, coalesced),
// then a (value, index) wave reduction picks the smallest distance, lowest index on ties (torch.argmin).
__global__ __launch_bounds__(256) void pq_encode_kernel(const float* x, int64_t ldx, const float* cen, const float* norm2,
                                                        int M, int dsub, int64_t n, uint8_t* codes) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* xr = x + row * ldx;
    for (int m = 0; m < M; ++m) {
        float best = INFINITY;
        int bi = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int c = lane + 64 * t;
            const float* cr = cen + ((int64_t)m * 256 + c) * dsub;
            float dot = 0.f;
            for (int e = 0; e < dsub; ++e) dot = fmaf(xr[m * dsub + e], cr[e], dot);
            const float dis = norm2[m * 256 + c] - 2.f * dot;
            if (dis < best) { best = dis; bi = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) codes[row * M + m] = (uint8_t)bi;
    }
}

__device__ __forceinline__ int owner_of(int64_t row, int64_t n_store, int64_t per, int world, int self) {
    if (row < 0 || row >= n_store) return self;
    const int64_t o = row / per;
    return (int)(o < world ? o : world - 1);
}

__global__ __launch_bounds__(256) void bucket_count_kernel(const int64_t* rows, int64_t n, int64_t n_store, int64_t per,
                                                           int world, int self, unsigned long long* counts) {
    __shared__ unsigned int h[64];
    if (threadIdx.x < 64) h[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        atomicAdd(&h[owner_of(rows[i], n_store, per, world, self)], 1u);
    __syncthreads();
    if ((int)threadIdx.x < world && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// cursor[o] starts at the exclusive prefix of counts; every block reserves a range per owner
__global__ __launch_bounds__(256) void bucket_scatter_kernel(const int64_t* rows, int64_t n, int64_t n_store, int64_t per,
                                                             int world, int self, unsigned long long* cursor,
                                                             int64_t* send_rows, int32_t* inv) {
    __shared__ unsigned int h[64];
    __shared__ unsigned long long base[64];
    const int64_t per_block = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per_block, hi = min(n, lo + per_block);
    if (threadIdx.x < 64) h[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) atomicAdd(&h[owner_of(rows[i], n_store, per, world, self)], 1u);
    __syncthreads();
    if ((int)threadIdx.x < world) {
        base[threadIdx.x] = h[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], (unsigned long long)h[threadIdx.x]) : 0ull;
        h[threadIdx.x] = 0;
    }
    __syncthreads();
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const int64_t row = rows[i];
        const int o = owner_of(row, n_store, per, world, self);
        const unsigned long long pos = base[o] + atomicAdd(&h[o], 1u);
        send_rows[pos] = row;
        inv[i] = (int32_t)pos;
    }
}

// Fixed-capacity variant for the sync-free exchange: owner o's requests go to send_rows[o * cap .. o * cap + cap), the
// unused tail keeps its -1 prefill, nothing depends on the bucket sizes on the host.  Rows that are not rows of the
// store are not sent at all; a request that does not fit its bucket is counted in *overflow.  Both get the index
// world * cap (the all-zero row the requester appends to the returned payload).
__global__ __launch_bounds__(256) void bucket_scatter_padded_kernel(const int64_t* rows, int64_t n, int64_t n_store, int64_t per,
                                                                    int world, int64_t cap, unsigned long long* cursor,
                                                                    int64_t* send_rows, int32_t* inv, unsigned long long* overflow) {
    __shared__ unsigned int h[64];
    __shared__ unsigned long long base[64];
    const int64_t per_block = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per_block, hi = min(n, lo + per_block);
    if (threadIdx.x < 64) h[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const int64_t row = rows[i];
        if (row >= 0 && row < n_store) atomicAdd(&h[owner_of(row, n_store, per, world, 0)], 1u);
    }
    __syncthreads();
    if ((int)threadIdx.x < world) {
        base[threadIdx.x] = h[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], (unsigned long long)h[threadIdx.x]) : 0ull;
        h[threadIdx.x] = 0;
    }
    __syncthreads();
    unsigned lost = 0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const int64_t row = rows[i];
        int32_t where = (int32_t)(world * cap);
        if (row >= 0 && row < n_store) {
            const int o = owner_of(row, n_store, per, world, 0);
            const unsigned long long pos = base[o] + atomicAdd(&h[o], 1u);
            if ((int64_t)pos < cap) {
                send_rows[o * cap + (int64_t)pos] = row;
                where = (int32_t)(o * cap + (int64_t)pos);
            } else {
                ++lost;
            }
        }
        inv[i] = where;
    }
    if (lost) atomicAdd(overflow, (unsigned long long)lost);
}

__global__ void bucket_prefix_kernel(const unsigned long long* counts, unsigned long long* cursor, int world) {
    if (threadIdx.x == 0) {
        unsigned long long acc = 0;
        for (int o = 0; o < world; ++o) { cursor[o] = acc; acc += counts[o]; }
    }
}

}  // namespace

int bucket_rows(const int64_t* rows, int64_t n, int64_t n_store, int64_t per, int world, int self, int64_t* counts,
                int64_t* cursor, int64_t* send_rows, int32_t* inv, hipStream_t stream) {
    GNNLM_REQUIRE(world >= 1 && world <= 64 && per > 0 && self >= 0 && self < world && n >= 0 && n < (1ll << 31), "bucket_rows: bad arguments");
    if (n == 0) return OK;
    GNNLM_REQUIRE(rows && counts && cursor && send_rows && inv, "bucket_rows: null operand");
    const unsigned blocks = (unsigned)std::min<int64_t>(cdiv(n, 1024), 1024);
    hipLaunchKernelGGL(bucket_count_kernel, dim3(blocks), dim3(256), 0, stream, rows, n, n_store, per, world, self,
                       reinterpret_cast<unsigned long long*>(counts));
    hipLaunchKernelGGL(bucket_prefix_kernel, dim3(1), dim3(64), 0, stream, reinterpret_cast<const unsigned long long*>(counts),
                       reinterpret_cast<unsigned long long*>(cursor), world);
    hipLaunchKernelGGL(bucket_scatter_kernel, dim3(blocks), dim3(256), 0, stream, rows, n, n_store, per, world, self,
                       reinterpret_cast<unsigned long long*>(cursor), send_rows, inv);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int bucket_rows_padded(const int64_t* rows, int64_t n, int64_t n_store, int64_t per, int world, int64_t cap, int64_t* cursor,
                       int64_t* send_rows, int32_t* inv, int64_t* overflow, hipStream_t stream) {
    GNNLM_REQUIRE(world >= 1 && world <= 64 && per > 0 && cap > 0 && n >= 0 && n < (1ll << 31) && world * cap < (1ll << 31) - 1,
                  "bucket_rows_padded: bad arguments");
    GNNLM_REQUIRE(cursor && send_rows && overflow, "bucket_rows_padded: null operand");
    GNNLM_HIP(hipMemsetAsync(cursor, 0, sizeof(int64_t) * world, stream));
    GNNLM_HIP(hipMemsetAsync(send_rows, 0xFF, sizeof(int64_t) * (size_t)(world * cap), stream));      // -1 everywhere
    if (n == 0) return OK;
    GNNLM_REQUIRE(rows && inv, "bucket_rows_padded: null operand");
    const unsigned blocks = (unsigned)std::min<int64_t>(cdiv(n, 1024), 1024);
    hipLaunchKernelGGL(bucket_scatter_padded_kernel, dim3(blocks), dim3(256), 0, stream, rows, n, n_store, per, world, cap,
                       reinterpret_cast<unsigned long long*>(cursor), send_rows, inv, reinterpret_cast<unsigned long long*>(overflow));
    GNNLM_LAUNCH_CHECK();
    return OK;
}

namespace {
// one request per group of LPR lanes (16 B each; LPR = row_bytes / 16), or one lane per request for short rows
__global__ __launch_bounds__(256) void gather_rows_peer_kernel(gnnlm_peer_gather_t d, int lpr) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t s = gid / lpr;
    const int part = (int)(gid - s * lpr);
    if (s >= d.n) return;
    const int64_t row = d.rows[s];
    bool ok = row >= 0 && row < d.n_store;
    int g = 0;
    int64_t local = 0;
    if (ok) {
        g = (int)min((int64_t)d.world - 1, row / d.rows_per_rank);
        local = row - d.shard_row0[g];
        ok = local >= 0 && local < d.shard_rows[g];
    }
    if (part == 0 && d.out_valid) d.out_valid[s] = ok ? 1 : 0;
    if (d.row_bytes >= 16) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ok) v = reinterpret_cast<const uint4*>(static_cast<const uint8_t*>(d.shard[g]) + local * d.row_bytes)[part];
        reinterpret_cast<uint4*>(static_cast<uint8_t*>(d.out) + s * d.row_bytes)[part] = v;
    } else {
        const uint8_t* src = static_cast<const uint8_t*>(d.shard[g]) + local * d.row_bytes;
        uint8_t* dst = static_cast<uint8_t*>(d.out) + s * d.row_bytes;
        for (int b = 0; b < d.row_bytes; ++b) dst[b] = ok ? src[b] : 0;
    }
}
}  // namespace

int gather_rows_peer(const gnnlm_peer_gather_t& d, hipStream_t stream) {
    GNNLM_REQUIRE(d.world >= 1 && d.world <= 16 && d.rows_per_rank > 0 && d.n_store > 0 && d.n >= 0 && d.n < (1ll << 31),
                  "gather_rows_peer: bad arguments");
    GNNLM_REQUIRE(d.row_bytes > 0 && (d.row_bytes <= 16 || d.row_bytes % 16 == 0) && d.row_bytes <= 4096, "gather_rows_peer: row_bytes must be 1..16 or a multiple of 16");
    if (d.n == 0) return OK;
    GNNLM_REQUIRE(d.rows && d.out, "gather_rows_peer: null operand");
    for (int g = 0; g < d.world; ++g)
        GNNLM_REQUIRE(d.shard_rows[g] == 0 || (d.shard[g] && (d.row_bytes < 16 || (uintptr_t)d.shard[g] % 16 == 0)), "gather_rows_peer: null / misaligned shard");
    GNNLM_REQUIRE(d.row_bytes < 16 || (uintptr_t)d.out % 16 == 0, "gather_rows_peer: misaligned output");
    const int lpr = d.row_bytes >= 16 ? d.row_bytes / 16 : 1;
    hipLaunchKernelGGL(gather_rows_peer_kernel, dim3((unsigned)cdiv(d.n * lpr, 256)), dim3(256), 0, stream, d, lpr);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int pq_encode(const float* x, int64_t ldx, const float* cen, const float* norm2, int M, int dsub, int64_t n, uint8_t* codes,
              hipStream_t stream) {
    GNNLM_REQUIRE(M > 0 && dsub > 0 && n >= 0, "pq_encode: bad shape");
    if (n == 0) return OK;
    GNNLM_REQUIRE(x && cen && norm2 && codes, "pq_encode: null operand");
    hipLaunchKernelGGL(pq_encode_kernel, dim3((unsigned)cdiv(n, 4)), dim3(256), 0, stream, x, ldx, cen, norm2, M, dsub, n, codes);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int gather_decode(const GatherParams& p, hipStream_t stream) {
    GNNLM_REQUIRE(p.left >= 0 && p.right >= 0 && p.n_groups >= 0, "gather_decode: bad shape");
    // an empty request (a rank of the sharded exchange that receives no rows) or an empty shard is legal and must
    // not fail on the null pointers torch hands out for empty tensors: its peers are already inside the next collective
    if (p.n_groups == 0) return OK;
    const bool mapped = p.shards != nullptr;    // the code rows come from a set of mapped shards (ABI 4)
    GNNLM_REQUIRE(!mapped || (!p.direct && !(p.out_labels && p.vals)), "gather_decode: a shard table excludes direct codes and a label table");
    GNNLM_REQUIRE((mapped || p.codes || p.n_local == 0) && (p.direct ? p.in_valid != nullptr : p.ids != nullptr), "gather_decode: null codes/ids");
    GNNLM_REQUIRE(p.M > 0 && p.dsub > 0 && p.dsub % 4 == 0, "gather_decode: dsub must be a multiple of 4");
    GNNLM_REQUIRE(!p.out_x || (p.centroids && p.ld_x % 4 == 0 && (uintptr_t)p.out_x % 16 == 0),
                  "gather_decode: out_x needs centroids, 16-byte alignment and ld % 4 == 0");
    GNNLM_REQUIRE(p.vals_itemsize == 2 || p.vals_itemsize == 4, "gather_decode: vals must be int16 or int32");
    const int64_t n_slots = p.n_groups * (1 + p.left + p.right);
    if (n_slots == 0) return OK;
    if (!mapped && !p.direct && !p.out_x && p.out_codes && p.left == 0 && p.right == 0 && p.M % 16 == 0 &&
        (uintptr_t)p.codes % 16 == 0 && (uintptr_t)p.out_codes % 16 == 0) {
        ProfScope prof(K_GATHER, stream, 0.0, (2.0 * p.M + 8.0) * n_slots);
        const int64_t blocks = std::min<int64_t>(cdiv(n_slots * (p.M >> 4), 256), 256 * 16);
        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p);
        GNNLM_LAUNCH_CHECK();
        return OK;
    }
    const int64_t blocks = std::min<int64_t>(cdiv(n_slots, 4), 256 * 16);
    const double row_bytes = (double)p.M + (p.out_x ? 4.0 * p.M * p.dsub : 0.0) + (p.out_codes ? p.M : 0.0) +
                             (p.out_labels ? 4.0 + p.vals_itemsize : 0.0) + (p.out_valid ? 1.0 : 0.0);
    ProfScope prof(K_GATHER, stream, 0.0, row_bytes * n_slots + 8.0 * p.n_groups);
    hipLaunchKernelGGL(gather_decode_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

}  // namespace gnnlm
