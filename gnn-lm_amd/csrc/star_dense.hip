// ('ntgt','inter','tgt') star attention over DENSE neighbour rows -- layers >= 1 of a multi-layer HGT, where a neighbour is
// the centre state of its context group after l ntgt updates (fairseq/models/hgt.py:354-356,383-385 with the neighbour-side
// projections absorbed into the query, see attn.hip): the rows come from the step's workspace or from the cross-batch
// centre-state cache.  HBM-bound by construction: 4 * D bytes per neighbour (4 KiB at d = 1024), 2.1 GB per layer for a
// 4096-token batch with k_g = 128 -- and nothing else worth mentioning.
//
// The generic kernel (attn.hip: star_attn_kernel) walks the rows TWICE (scores, then the weighted sum; a token's 512 KiB of rows
// do not survive in L2 between the passes: 1.7 ms per layer of a 4096-token batch = 1.2 TB/s of algorithmic bytes).  This one
// reads every row ONCE: each wave owns whole rows (lane l holds floats [4 QPL l, 4 QPL (l + 1)) of the row), two rows per trip,
// scores for the 8 heads by a transposing butterfly (reduce8), and a running softmax per head (max, normaliser and weighted sum
// rescaled when -- rarely -- the max moves), so the row is still in registers when its weight is known.  The query rows
// U[i, h, :] live in LDS (32 KiB), read as ds_read_b128 against two rows at a time.  The four waves keep independent softmax
// states over their quarter of the neighbours and are merged at the end in a fixed order (deterministic: same bits every run,
// and for every way -- merged groups, cache slots, un-merged workspace rows -- the same neighbour rows reach the token).
#include "kernels.h"

namespace gnnlm {
namespace {

constexpr int HB = 8;

// lane 8 h ends up with the sum of v[h] over the wave (same butterfly as attn.hip)
__device__ __forceinline__ float reduce8(const float (&v)[HB], int lane) {
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
    float w[4], y[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float send = b5 ? v[t] : v[t + 4];
        const float keep = b5 ? v[t + 4] : v[t];
        w[t] = keep + __shfl_xor(send, 32, 64);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float send = b4 ? w[t] : w[t + 2];
        const float keep = b4 ? w[t + 2] : w[t];
        y[t] = keep + __shfl_xor(send, 16, 64);
    }
    float r;
    {
        const float send = b3 ? y[0] : y[1];
        const float keep = b3 ? y[1] : y[0];
        r = keep + __shfl_xor(send, 8, 64);
    }
    r += __shfl_xor(r, 4, 64);
    r += __shfl_xor(r, 2, 64);
    r += __shfl_xor(r, 1, 64);
    return r;
}

// a wave-uniform value the compiler cannot see as one (read from LDS at a uniform address): into SGPRs
__device__ __forceinline__ int64_t uniform64(int64_t v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffff));
    const int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
    return ((int64_t)hi << 32) | lo;
}
__device__ __forceinline__ float uniformf(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

template <int QPL>
__global__ __launch_bounds__(256, 2) void star_dense_kernel(StarAttnParams p) {
    constexpr int D = 256 * QPL, NQ = D / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* us = smem;                                           // [HB][D] query rows, then the merged Z
    float* red = us + HB * D;                                   // [4 waves][HB][2] (max, normaliser) of each wave
    int64_t* xrow = reinterpret_cast<int64_t*>(red + 4 * HB * 2);   // [kg] row of neighbour j in X, -1: not a neighbour
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = blockIdx.x, kg = p.kg;
    const int64_t* ids = p.ids + (int64_t)i * kg;

    for (int e = tid; e < HB * NQ; e += 256) {
        const int h = e / NQ, q = e - h * NQ;
        reinterpret_cast<float4*>(us)[e] = h < p.H ? *reinterpret_cast<const float4*>(p.U + ((int64_t)i * p.H + h) * D + 4 * q)
                                                   : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int j = tid; j < kg; j += 256) xrow[j] = star_nb_ok(p, i, j, ids[j]) ? star_group(p, i, j) * p.x_group_stride : -1;
    __syncthreads();

    float m[HB], l[HB];
    float4 z[HB][QPL];
#pragma unroll
    for (int h = 0; h < HB; ++h) {
        m[h] = -INFINITY;
        l[h] = 0.f;
#pragma unroll
        for (int t = 0; t < QPL; ++t) z[h][t] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // two rows per trip and wave: neighbours 8 c + 2 w, 8 c + 2 w + 1
    for (int j0 = 2 * wave; j0 < kg; j0 += 8) {
        const int64_t r0 = uniform64(xrow[j0]), r1 = j0 + 1 < kg ? uniform64(xrow[j0 + 1]) : -1;
        if (r0 < 0 && r1 < 0) continue;                           // (uniform)
        float4 xa[QPL], xb[QPL];
#pragma unroll
        for (int t = 0; t < QPL; ++t) {
            xa[t] = r0 >= 0 ? *reinterpret_cast<const float4*>(p.X + r0 * p.ldx + 4 * (QPL * lane + t)) : make_float4(0.f, 0.f, 0.f, 0.f);
            xb[t] = r1 >= 0 ? *reinterpret_cast<const float4*>(p.X + r1 * p.ldx + 4 * (QPL * lane + t)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float pa[HB], pb[HB];
#pragma unroll
        for (int h = 0; h < HB; ++h) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int t = 0; t < QPL; ++t) {
                const float4 u = *reinterpret_cast<const float4*>(us + h * D + 4 * (QPL * lane + t));
                a = fmaf(xa[t].x, u.x, a); b = fmaf(xb[t].x, u.x, b);
                a = fmaf(xa[t].y, u.y, a); b = fmaf(xb[t].y, u.y, b);
                a = fmaf(xa[t].z, u.z, a); b = fmaf(xb[t].z, u.z, b);
                a = fmaf(xa[t].w, u.w, a); b = fmaf(xb[t].w, u.w, b);
            }
            pa[h] = a;
            pb[h] = b;
        }
        const float ta = reduce8(pa, lane), tb = reduce8(pb, lane);
#pragma unroll
        for (int h = 0; h < HB; ++h) {
            // the two scores of head h, wave-uniform (v_readlane: no LDS round trip)
            const float sa = r0 >= 0 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ta), 8 * h)) : -INFINITY;
            const float sb = r1 >= 0 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tb), 8 * h)) : -INFINITY;
            const float mn = uniformf(fmaxf(m[h], fmaxf(sa, sb)));
            if (mn > m[h]) {                                      // the running max moved: rescale what was summed so far (uniform branch)
                const float sc = expf(m[h] - mn);                 // (m = -inf: 0)
                l[h] *= sc;
#pragma unroll
                for (int t = 0; t < QPL; ++t) { z[h][t].x *= sc; z[h][t].y *= sc; z[h][t].z *= sc; z[h][t].w *= sc; }
                m[h] = mn;
            }
            const float wa = r0 >= 0 ? expf(sa - mn) : 0.f, wb = r1 >= 0 ? expf(sb - mn) : 0.f;
            l[h] += wa + wb;
#pragma unroll
            for (int t = 0; t < QPL; ++t) {
                z[h][t].x = fmaf(wa, xa[t].x, z[h][t].x); z[h][t].x = fmaf(wb, xb[t].x, z[h][t].x);
                z[h][t].y = fmaf(wa, xa[t].y, z[h][t].y); z[h][t].y = fmaf(wb, xb[t].y, z[h][t].y);
                z[h][t].z = fmaf(wa, xa[t].z, z[h][t].z); z[h][t].z = fmaf(wb, xb[t].z, z[h][t].z);
                z[h][t].w = fmaf(wa, xa[t].w, z[h][t].w); z[h][t].w = fmaf(wb, xb[t].w, z[h][t].w);
            }
        }
    }
    // ---- merge the four waves (fixed order): M = max_w m_w, L = sum_w l_w e^(m_w - M), Z = sum_w z_w e^(m_w - M) / L
    if (lane < HB) {
        float mv = m[0], lv = l[0];
#pragma unroll
        for (int h = 1; h < HB; ++h)
            if (lane == h) { mv = m[h]; lv = l[h]; }
        red[(wave * HB + lane) * 2] = mv;
        red[(wave * HB + lane) * 2 + 1] = lv;
    }
    __syncthreads();                                             // (also: every wave is done reading the query rows in `us`)
    float f[HB];
    bool any = false;
#pragma unroll
    for (int h = 0; h < HB; ++h) {
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) M = fmaxf(M, red[(w * HB + h) * 2]);
        float L = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = red[(w * HB + h) * 2];
            L += mw == -INFINITY ? 0.f : red[(w * HB + h) * 2 + 1] * expf(mw - M);
        }
        f[h] = (L > 0.f && m[h] != -INFINITY) ? expf(m[h] - M) / L : 0.f;
        any = any || L > 0.f;
    }
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int h = 0; h < HB; ++h)
#pragma unroll
                for (int t = 0; t < QPL; ++t) {
                    float4* dst = reinterpret_cast<float4*>(us + h * D + 4 * (QPL * lane + t));
                    float4 c = w == 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : *dst;
                    c.x = fmaf(z[h][t].x, f[h], c.x); c.y = fmaf(z[h][t].y, f[h], c.y);
                    c.z = fmaf(z[h][t].z, f[h], c.z); c.w = fmaf(z[h][t].w, f[h], c.w);
                    *dst = c;
                }
        }
        __syncthreads();
    }
    for (int e = tid; e < HB * NQ; e += 256) {
        const int h = e / NQ, q = e - h * NQ;
        if (h < p.H) *reinterpret_cast<float4*>(p.Z + ((int64_t)i * p.H + h) * D + 4 * q) = reinterpret_cast<const float4*>(us)[e];
    }
    if (tid == 0 && p.has_nb) p.has_nb[i] = any ? 1.f : 0.f;
}

}  // namespace

static size_t star_dense_lds_bytes(const StarAttnParams& p) {
    return (size_t)(HB * p.D + 4 * HB * 2) * sizeof(float) + (size_t)p.kg * sizeof(int64_t);
}

// (the LDS bound is part of eligibility: a shape with a very large k_g keeps the generic two-pass kernel instead of failing here)
bool star_attn_dense_eligible(const StarAttnParams& p) {
    static const bool off = getenv("GNNLM_STAR_GENERIC") != nullptr;      // A/B runs: the two-pass kernel of attn.hip
    return !off && p.X && !p.codes && !p.shards && p.H <= HB && (p.D == 256 || p.D == 512 || p.D == 1024) && p.ldx % 4 == 0 &&
           (uintptr_t)p.X % 16 == 0 && (uintptr_t)p.U % 16 == 0 && (uintptr_t)p.Z % 16 == 0 && p.kg >= 0 && star_dense_lds_bytes(p) <= 64 * 1024;
}

int star_attn_dense(const StarAttnParams& p, hipStream_t stream) {
    GNNLM_REQUIRE(star_attn_dense_eligible(p), "star_attn_dense: shape not supported");
    const size_t lds = star_dense_lds_bytes(p);
    dim3 grid(p.T), block(256);
    if (p.D == 1024) hipLaunchKernelGGL(star_dense_kernel<4>, grid, block, lds, stream, p);
    else if (p.D == 512) hipLaunchKernelGGL(star_dense_kernel<2>, grid, block, lds, stream, p);
    else hipLaunchKernelGGL(star_dense_kernel<1>, grid, block, lds, stream, p);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

}  // namespace gnnlm
