// Row-wise kernels of the eval hot path: LayerNorm epilogue of HGTLayer (fairseq/models/hgt.py:397-407),
// the log-softmax reductions of the tied adaptive softmax (fairseq/modules/adaptive_softmax.py:184-203)
// and the kNN-LM distance-softmax interpolation (knn/knn_model.py:192-217,
// fairseq/sequence_scorer.py:55-68,110,121).  All HBM/latency-bound; f32 arithmetic like the reference.
#include <hip/hip_fp16.h>
#include "kernels.h"

namespace gnnlm {
namespace {

// One wave per row.  out = LayerNorm(x (+ residual)) * gamma + beta.  The optional residual is the `+ h` of
// `trans_out + h` (hgt.py:403): added here, in the same (alpha * acc + bias) + residual order the GEMM epilogue
// would use, it is a streaming read instead of 16 scattered loads per accumulator tile behind the GEMM's stores.
// REGS: the row lives in registers (d <= 1024, d % 4 == 0, 16-byte aligned rows): one pass over memory.
template <bool REGS>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, int64_t ldx, const float* residual, int64_t ldr,
                                                        const float* gamma, const float* beta, float* out, int64_t ldo,
                                                        int64_t rows, int d, float eps, const uint8_t* valid,
                                                        const int32_t* rows_idx, const int32_t* n_dev, int n_mult) {
    const int lane = threadIdx.x & 63;
    // (a device-side count -- rows = min(rows, *n_dev * n_mult), ABI 9 -- comes with a capped grid: the waves walk the rows)
    if (n_dev) rows = min(rows, (int64_t)*n_dev * n_mult);
  for (int64_t r_ = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r_ < rows; r_ += (int64_t)gridDim.x * 4) {
    const int64_t row = rows_idx ? (int64_t)rows_idx[r_] : r_;          // a subset of the rows of x / residual / out / valid
    const float* xr = x + row * ldx;
    const float* rr = residual ? residual + row * ldr : nullptr;
    float* orow = out + row * ldo;
    if (valid && !valid[row]) {
        for (int e = lane; e < d; e += 64) orow[e] = 0.f;
        continue;
    }
    if constexpr (REGS) {
        float4 v[4];
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = 4 * (lane + 64 * t);
            v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < d) {
                v[t] = *reinterpret_cast<const float4*>(xr + e);
                if (rr) {
                    const float4 r = *reinterpret_cast<const float4*>(rr + e);
                    v[t].x += r.x; v[t].y += r.y; v[t].z += r.z; v[t].w += r.w;
                }
                s += (v[t].x + v[t].y) + (v[t].z + v[t].w);
            }
        }
        const float mean = wave_sum(s) / d;
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (4 * (lane + 64 * t) < d) {
                const float a = v[t].x - mean, b = v[t].y - mean, c = v[t].z - mean, e_ = v[t].w - mean;
                q = fmaf(a, a, fmaf(b, b, fmaf(c, c, fmaf(e_, e_, q))));
            }
        const float rstd = rsqrtf(wave_sum(q) / d + eps);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = 4 * (lane + 64 * t);
            if (e < d) {
                const float4 g = *reinterpret_cast<const float4*>(gamma + e), bt = *reinterpret_cast<const float4*>(beta + e);
                float4 o;
                o.x = (v[t].x - mean) * rstd * g.x + bt.x; o.y = (v[t].y - mean) * rstd * g.y + bt.y;
                o.z = (v[t].z - mean) * rstd * g.z + bt.z; o.w = (v[t].w - mean) * rstd * g.w + bt.w;
                *reinterpret_cast<float4*>(orow + e) = o;
            }
        }
    } else {
        float s = 0.f;
        for (int e = lane; e < d; e += 64) s += xr[e] + (rr ? rr[e] : 0.f);
        const float mean = wave_sum(s) / d;
        float v = 0.f;
        for (int e = lane; e < d; e += 64) {
            const float c = xr[e] + (rr ? rr[e] : 0.f) - mean;
            v = fmaf(c, c, v);
        }
        const float rstd = rsqrtf(wave_sum(v) / d + eps);
        for (int e = lane; e < d; e += 64) orow[e] = (xr[e] + (rr ? rr[e] : 0.f) - mean) * rstd * gamma[e] + beta[e];
    }
  }
}

__global__ void mean2_kernel(const float* a, const float* b, float* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = 0.5f * (a[i] + b[i]);
}

__global__ void half_to_float_kernel(const __half* src, float* dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = __half2float(src[i]);
}

// out[i, j] = -1 where the neighbour lies inside its own token's context (|pos[i] - ids[i, j]| < ctx), else ids[i, j]
__global__ void filter_neighbors_kernel(const int64_t* ids, const int64_t* pos, int64_t n, int kg, int64_t ctx, int64_t* out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * kg) return;
    const int64_t id = ids[e], dlt = pos[e / kg] - id;
    out[e] = (id != -1 && (dlt < 0 ? -dlt : dlt) < ctx) ? (int64_t)-1 : id;
}

// one workgroup per row: lse = logsumexp(row[:n]), picked = row[pick]
__global__ __launch_bounds__(256) void row_lse_pick_kernel(const float* logits, int64_t ld, int64_t rows,
                                                           const int32_t* m_dev, int n, const int32_t* pick,
                                                           float* lse, float* picked) {
    __shared__ float red_m[4], red_s[4];
    const int64_t row = blockIdx.x;
    if (m_dev && row >= *m_dev) return;
    const float* r = logits + row * ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float m = -INFINITY, s = 0.f;
    for (int e = tid; e < n; e += 256) {
        const float v = r[e];
        if (v > m) {
            s = s * expf(m - v) + 1.f;
            m = v;
        } else {
            s += expf(v - m);
        }
    }
    // combine (m, s) pairs across the wave, then across the 4 waves
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        const float mm = fmaxf(m, m2);
        s = (m == -INFINITY ? 0.f : s * expf(m - mm)) + (m2 == -INFINITY ? 0.f : s2 * expf(m2 - mm));
        m = mm;
    }
    if (lane == 0) { red_m[wave] = m; red_s[wave] = s; }
    __syncthreads();
    if (tid == 0) {
        float mm = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
        float ss = 0.f;
        for (int w = 0; w < 4; ++w) ss += red_m[w] == -INFINITY ? 0.f : red_s[w] * expf(red_m[w] - mm);
        lse[row] = mm + logf(ss);
        if (picked) picked[row] = pick ? r[pick[row]] : 0.f;
    }
}

// one workgroup of 16 waves: stable compaction of the rows whose target falls into each tail band.  Rows are
// taken in chunks of 1024; per band a ballot per wave + a 16-entry LDS prefix gives every row its slot.
constexpr int BS_WAVES = 16;
__global__ __launch_bounds__(64 * BS_WAVES) void band_split_kernel(BandSplitParams p) {
    __shared__ int wcnt[8][BS_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int count[8] = {0};
    for (int64_t base = 0; base < p.n; base += 64 * BS_WAVES) {
        const int64_t r = base + tid;
        const bool in = r < p.n;
        const int64_t t = in ? p.target[r] : 0;
        int band = 0;
#pragma unroll
        for (int b = 1; b < 8; ++b)
            if (b < p.n_bands && t >= p.cutoff[b - 1]) band = b;
        if (in) p.head_pick[r] = band == 0 ? (int32_t)t : p.cutoff[0] + band - 1;
        unsigned long long mask[8];
#pragma unroll
        for (int b = 1; b < 8; ++b) {
            mask[b] = 0;
            if (b < p.n_bands) {
                mask[b] = __ballot(in && band == b);
                if (lane == 0) wcnt[b][wave] = __popcll(mask[b]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int b = 1; b < 8; ++b) {
            if (b >= p.n_bands) break;
            int before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < BS_WAVES; ++w) {
                before += w < wave ? wcnt[b][w] : 0;
                total += wcnt[b][w];
            }
            if (in && band == b) {
                const int64_t dst = (int64_t)(b - 1) * p.n + count[b] + before + __popcll(mask[b] & ((1ull << lane) - 1ull));
                p.band_rows[dst] = (int32_t)r;
                p.band_pick[dst] = (int32_t)(t - p.cutoff[b - 1]);
            }
            count[b] += total;
        }
        __syncthreads();
    }
    if (tid == 0)
        for (int b = 1; b < p.n_bands; ++b) p.band_count[b - 1] = count[b];
}

__global__ void head_logp_kernel(const float* picked, const float* lse, float* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = picked[i] - lse[i];
}

__global__ void tail_combine_kernel(const float* tail_picked, const float* tail_lse, const int32_t* rows,
                                    const int32_t* count_dev, int64_t n_max, float* lm_logp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = count_dev ? min((int64_t)*count_dev, n_max) : n_max;
    if (i < n) {
        const int64_t r = rows ? rows[i] : i;
        lm_logp[r] += tail_picked[i] - tail_lse[i];
    }
}

// tag of a label (gnnlm_knn_interp_t.vals_tag): the top byte of a multiplicative hash of its low 32 bits
__device__ __forceinline__ uint32_t label_tag(int64_t v) { return ((uint32_t)v * 2654435761u) >> 24; }
__global__ __launch_bounds__(256) void label_tags_kernel(const void* vals, int itemsize, int64_t n, uint8_t* tag) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const int64_t v = itemsize == 2 ? (int64_t) reinterpret_cast<const int16_t*>(vals)[r] : (int64_t) reinterpret_cast<const int32_t*>(vals)[r];
    tag[r] = (uint8_t)label_tag(v);
}

// one wave per token
__global__ __launch_bounds__(256) void knn_interp_kernel(KnnInterpParams p, float log_1ml, float log_l) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= p.n) return;
    const float* sims = p.sims + i * p.k;
    const int64_t* ids = p.ids + i * p.k;
    const int64_t tgt = p.targets[i];
    float mx = -INFINITY;
    for (int j = lane; j < p.k; j += 64) {
        const float s = (ids[j] == -1 ? -1e10f : sims[j]) / p.temperature;      // knn_model.py:193,196
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float den = 0.f, num = 0.f;
    int rec = 0;
    for (int j = lane; j < p.k; j += 64) {
        const int64_t id = ids[j];
        const float s = (id == -1 ? -1e10f : sims[j]) / p.temperature;
        const float e = expf(s - mx);
        int64_t val;
        if (p.knn_vals) {
            val = p.knn_vals[i * p.k + j];
        } else {
            // numpy indexing semantics of vals[knns] (knn_model.py:198): -1 wraps to the last row
            const int64_t row = (id < 0 ? id + p.n_store : id) - p.row0;
            val = -1;                                       // rows outside the shard / the store never match a target
            if (row >= 0 && row < p.n_local)
                val = p.vals_itemsize == 2 ? (int64_t) reinterpret_cast<const int16_t*>(p.vals)[row]
                                           : (int64_t) reinterpret_cast<const int32_t*>(p.vals)[row];
        }
        const bool hit = val == tgt;                                             // :211
        den += e;
        num += hit ? e : 0.f;
        rec += hit;
    }
    den = wave_sum(den);
    num = wave_sum(num);
    rec = (int)wave_sum((float)rec);
    if (lane == 0) {
        const float pk = num / den;
        if (p.out_pknn) p.out_pknn[i] = pk;
        if (p.out_recall) p.out_recall[i] = rec;
        // sequence_scorer.py:55-68 with knn_probs = log(p + 1e-10) (:121)
        const float a = p.lm_logp[i] + log_1ml;
        const float b = logf(pk + 1e-10f) + log_l;
        const float m = fmaxf(a, b);
        p.out_logp[i] = m + logf(expf(a - m) + expf(b - m));
    }
}

// k <= 64 * JMAX: every lane keeps its JMAX (id, sim) pairs in registers and has all its label gathers
// (random 4-B reads of the 413-MB table, the kernel's cost) in flight at once; one pass.  Same summation
// order as the generic kernel above.
template <int JMAX>
__global__ __launch_bounds__(256) void knn_interp_regs_kernel(KnnInterpParams p, float log_1ml, float log_l) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= p.n) return;
    const float* sims = p.sims + i * p.k;
    const int64_t* ids = p.ids + i * p.k;
    int64_t id[JMAX];
    float sv[JMAX];
    int64_t val[JMAX];
#pragma unroll
    for (int t = 0; t < JMAX; ++t) {
        const int j = lane + 64 * t;
        id[t] = j < p.k ? ids[j] : -1;
        sv[t] = j < p.k ? sims[j] : 0.f;
    }
    const int64_t tgt = p.targets[i];
    if (p.vals_tag && !p.knn_vals) {
        // tag table: k random 1-byte reads; the 4-byte label is read only where the tag equals the target's (rare)
        const uint32_t ttag = label_tag(tgt);
        uint32_t tg[JMAX];
#pragma unroll
        for (int t = 0; t < JMAX; ++t) {
            const int j = lane + 64 * t;
            const int64_t row = (id[t] < 0 ? id[t] + p.n_store : id[t]) - p.row0;
            tg[t] = 0x100u;                                 // no tag: rows outside the shard / the store never match
            if (j < p.k && row >= 0 && row < p.n_local) tg[t] = p.vals_tag[row];
        }
#pragma unroll
        for (int t = 0; t < JMAX; ++t) {
            val[t] = ~tgt;                                  // "no match": equal labels have equal tags
            if (tg[t] == ttag) {
                const int64_t row = (id[t] < 0 ? id[t] + p.n_store : id[t]) - p.row0;
                val[t] = p.vals_itemsize == 2 ? (int64_t) reinterpret_cast<const int16_t*>(p.vals)[row]
                                              : (int64_t) reinterpret_cast<const int32_t*>(p.vals)[row];
            }
        }
    } else
#pragma unroll
    for (int t = 0; t < JMAX; ++t) {
        const int j = lane + 64 * t;
        val[t] = -1;
        if (j < p.k) {
            if (p.knn_vals) {
                val[t] = p.knn_vals[i * p.k + j];
            } else {
                const int64_t row = (id[t] < 0 ? id[t] + p.n_store : id[t]) - p.row0;
                if (row >= 0 && row < p.n_local)
                    val[t] = p.vals_itemsize == 2 ? (int64_t) reinterpret_cast<const int16_t*>(p.vals)[row]
                                                  : (int64_t) reinterpret_cast<const int32_t*>(p.vals)[row];
            }
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < JMAX; ++t) {
        sv[t] = (id[t] == -1 ? -1e10f : sv[t]) / p.temperature;
        if (lane + 64 * t < p.k) mx = fmaxf(mx, sv[t]);
    }
    mx = wave_max(mx);
    float den = 0.f, num = 0.f;
    int rec = 0;
#pragma unroll
    for (int t = 0; t < JMAX; ++t) {
        if (lane + 64 * t < p.k) {
            const float e = expf(sv[t] - mx);
            const bool hit = val[t] == tgt;
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
        const float b = logf(pk + 1e-10f) + log_l;
        const float m = fmaxf(a, b);
        p.out_logp[i] = m + logf(expf(a - m) + expf(b - m));
    }
}

__global__ __launch_bounds__(256) void gelu_kernel(float* x, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const float v = x[i];
        x[i] = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    }
}

__global__ __launch_bounds__(1024) void masked_sum_f64_kernel(const float* x, const uint8_t* mask, int64_t n, double* out) {
    __shared__ double red[1024];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) s += (!mask || mask[i]) ? (double)x[i] : 0.0;
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] += red[0];
}

}  // namespace

// idx[g * n_sel + j] = g * n_g + sel[j]: the rows of the slots `sel` of every group
struct SlotSel { int c[8]; };
__global__ __launch_bounds__(256) void group_rows_kernel(int32_t* idx, int64_t n_groups, int n_g, SlotSel sel, int n_sel) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_groups * n_sel) return;
    const int64_t g = e / n_sel;
    idx[e] = (int32_t)(g * n_g + sel.c[(int)(e - g * n_sel)]);
}
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ src, int64_t ld_src, float* __restrict__ dst, int64_t ld_dst,
                                                           const int32_t* __restrict__ slots, int64_t n, int d4, const int32_t* n_dev) {
    if (n_dev) n = min(n, (int64_t)*n_dev);
    for (int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); g < n; g += (int64_t)gridDim.x * 4) {      // one wave per row
        const int64_t slot = slots[g];
        if (slot < 0) continue;
        const float4* a = reinterpret_cast<const float4*>(src + g * ld_src);
        float4* o = reinterpret_cast<float4*>(dst + slot * ld_dst);
        for (int e = threadIdx.x & 63; e < d4; e += 64) o[e] = a[e];
    }
}
int scatter_rows(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, const int32_t* slots, int64_t n, int d, hipStream_t stream,
                 const int32_t* n_dev) {
    GNNLM_REQUIRE(src && dst && slots && n >= 0 && d > 0 && d % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0, "scatter_rows: bad arguments");
    if (n == 0) return OK;
    const int64_t wgs = cdiv(n, (int64_t)4);
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)(n_dev ? std::min<int64_t>(wgs, 256 * 16) : wgs)), dim3(256), 0, stream, src, ld_src, dst, ld_dst,
                       slots, n, d / 4, n_dev);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int group_rows(int32_t* idx, int64_t n_groups, int n_g, const int* sel, int n_sel, hipStream_t stream) {
    GNNLM_REQUIRE(idx && n_sel > 0 && n_sel <= 8 && n_groups * n_g < (1ll << 31), "group_rows: bad arguments");
    if (n_groups == 0) return OK;
    SlotSel s{};
    for (int j = 0; j < n_sel; ++j) s.c[j] = sel[j];
    hipLaunchKernelGGL(group_rows_kernel, dim3((unsigned)cdiv(n_groups * n_sel, 256)), dim3(256), 0, stream, idx, n_groups, n_g, s, n_sel);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int layernorm(const float* x, int64_t ldx, const float* gamma, const float* beta, float* out, int64_t ldo,
              int64_t rows, int d, float eps, const uint8_t* valid, hipStream_t stream, const float* residual, int64_t ldr,
              const int32_t* rows_idx, const int32_t* n_dev, int n_mult) {
    GNNLM_REQUIRE(x && gamma && beta && out && d > 0, "layernorm: bad arguments");
    if (rows == 0) return OK;
    ProfScope prof(K_LAYERNORM, stream, 0.0, (residual ? 12.0 : 8.0) * rows * d);
    const bool a16 = ((uintptr_t)x | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)residual) % 16 == 0;
    const bool regs = d <= 1024 && d % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && (!residual || ldr % 4 == 0) && a16;
    const dim3 grid((unsigned)(n_dev ? std::min<int64_t>(cdiv(rows, 4), 256 * 32) : cdiv(rows, 4))), block(256);
    if (regs) hipLaunchKernelGGL(layernorm_kernel<true>, grid, block, 0, stream, x, ldx, residual, ldr, gamma, beta, out, ldo, rows, d, eps, valid, rows_idx, n_dev, n_mult);
    else hipLaunchKernelGGL(layernorm_kernel<false>, grid, block, 0, stream, x, ldx, residual, ldr, gamma, beta, out, ldo, rows, d, eps, valid, rows_idx, n_dev, n_mult);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int mean2(const float* a, const float* b, float* out, int64_t n, hipStream_t stream) {
    if (n == 0) return OK;
    hipLaunchKernelGGL(mean2_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream, a, b, out, n);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int filter_neighbors(const int64_t* ids, const int64_t* pos, int64_t n, int kg, int64_t ctx, int64_t* out, hipStream_t stream) {
    GNNLM_REQUIRE(n >= 0 && kg > 0 && ctx >= 0, "filter_neighbors: bad shape");
    if (n == 0) return OK;
    GNNLM_REQUIRE(ids && pos && out, "filter_neighbors: null");
    GNNLM_REQUIRE(cdiv(n * kg, (int64_t)256) < (1ll << 31), "filter_neighbors: too many ids for one launch");
    hipLaunchKernelGGL(filter_neighbors_kernel, dim3((unsigned)cdiv(n * kg, (int64_t)256)), dim3(256), 0, stream, ids, pos, n, kg, ctx, out);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int half_to_float(const void* src, float* dst, int64_t n, hipStream_t stream) {
    GNNLM_REQUIRE(src && dst, "half_to_float: null");
    if (n == 0) return OK;
    hipLaunchKernelGGL(half_to_float_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream,
                       reinterpret_cast<const __half*>(src), dst, n);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int row_lse_pick(const float* logits, int64_t ld, int64_t rows, const int32_t* m_dev, int n, const int32_t* pick,
                 float* lse, float* picked, hipStream_t stream) {
    GNNLM_REQUIRE(logits && lse && n > 0, "row_lse_pick: bad arguments");
    if (rows == 0) return OK;
    ProfScope prof(K_LSE, stream, 0.0, 4.0 * rows * n, m_dev, (double)rows);
    hipLaunchKernelGGL(row_lse_pick_kernel, dim3((unsigned)rows), dim3(256), 0, stream, logits, ld, rows, m_dev, n,
                       pick, lse, picked);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int band_split(const BandSplitParams& p, hipStream_t stream) {
    GNNLM_REQUIRE(p.target && p.head_pick && p.n_bands >= 1 && p.n_bands <= 8, "band_split: bad arguments");
    GNNLM_REQUIRE(p.n_bands == 1 || (p.band_rows && p.band_pick && p.band_count), "band_split: null band outputs");
    hipLaunchKernelGGL(band_split_kernel, dim3(1), dim3(64 * BS_WAVES), 0, stream, p);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int head_logp(const float* picked, const float* lse, float* out, int64_t n, hipStream_t stream) {
    if (n == 0) return OK;
    hipLaunchKernelGGL(head_logp_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream, picked, lse, out, n);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int tail_combine(const float* tail_picked, const float* tail_lse, const int32_t* rows, const int32_t* count_dev,
                 int64_t n_max, float* lm_logp, hipStream_t stream) {
    if (n_max == 0) return OK;
    hipLaunchKernelGGL(tail_combine_kernel, dim3((unsigned)cdiv(n_max, 256)), dim3(256), 0, stream, tail_picked,
                       tail_lse, rows, count_dev, n_max, lm_logp);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int knn_interp(const KnnInterpParams& p, hipStream_t stream) {
    GNNLM_REQUIRE(p.lm_logp && p.sims && p.ids && p.targets && p.out_logp, "knn_interp: null operand");
    GNNLM_REQUIRE(p.knn_vals || p.vals, "knn_interp: need vals or pre-fetched knn_vals");
    GNNLM_REQUIRE(p.knn_vals || (p.n_local > 0 && p.n_store > 0 && p.row0 >= 0), "knn_interp: vals needs n_store / row0 / n_local");
    GNNLM_REQUIRE(p.k > 0 && p.temperature > 0.f && p.lmbda > 0.0 && p.lmbda < 1.0, "knn_interp: need k>0, t>0, 0<lmbda<1");
    GNNLM_REQUIRE(p.vals_itemsize == 2 || p.vals_itemsize == 4, "knn_interp: vals must be int16 or int32");
    if (p.n == 0) return OK;
    // coefficients are float32 roundings of the float64 logs, as in coeffs[0] = np.log(1 - coeff)
    ProfScope prof(K_KNN, stream, 0.0, (double)p.n * p.k * (12.0 + (p.knn_vals ? 4.0 : p.vals_itemsize)) + 16.0 * p.n);
    const float log_1ml = (float)log(1.0 - p.lmbda), log_l = (float)log(p.lmbda);
    if (knn_interp_bucketed_eligible(p)) return knn_interp_bucketed(p, log_1ml, log_l, stream);      // routed look-ups (knn_bucket.hip)
    const dim3 grid((unsigned)cdiv(p.n, 4)), block(256);
    if (p.k <= 256) hipLaunchKernelGGL(knn_interp_regs_kernel<4>, grid, block, 0, stream, p, log_1ml, log_l);
    else if (p.k <= 1024) hipLaunchKernelGGL(knn_interp_regs_kernel<16>, grid, block, 0, stream, p, log_1ml, log_l);
    else hipLaunchKernelGGL(knn_interp_kernel, grid, block, 0, stream, p, log_1ml, log_l);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int label_tags(const void* vals, int itemsize, int64_t n, uint8_t* tag, hipStream_t stream) {
    GNNLM_REQUIRE(vals && tag && n >= 0 && (itemsize == 2 || itemsize == 4), "label_tags: need an int16 / int32 label table");
    if (n == 0) return OK;
    GNNLM_REQUIRE(cdiv(n, (int64_t)256) < (1ll << 31), "label_tags: too many rows for one launch");
    hipLaunchKernelGGL(label_tags_kernel, dim3((unsigned)cdiv(n, (int64_t)256)), dim3(256), 0, stream, vals, itemsize, n, tag);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int gelu(float* x, int64_t n, hipStream_t stream) {
    GNNLM_REQUIRE(n >= 0, "gelu: bad size");
    if (n == 0) return OK;
    GNNLM_REQUIRE(x, "gelu: null");
    hipLaunchKernelGGL(gelu_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream, x, n);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int masked_sum_f64(const float* x, const uint8_t* mask, int64_t n, double* out, hipStream_t stream) {
    GNNLM_REQUIRE(x && out, "masked_sum_f64: null");
    hipLaunchKernelGGL(masked_sum_f64_kernel, dim3(1), dim3(1024), 0, stream, x, mask, n, out);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

}  // namespace gnnlm
