// HGT attention kernels (reference: HGTLayer.forward, fairseq/models/hgt.py:341-386).
//
// The three edge types of the token/neighbour graph (fairseq/models/transformer.py:913-918) each get
// the kernel their shape asks for -- the graph is implicit, no edge list is ever built:
//   ('ntgt','inter','tgt')   star_attn   : every token attends over its kg neighbour centres.  The
//        per-neighbour K/V projections are absorbed into the query side (exact algebra):
//          score = q.(x W_k' ) = x.(W_k' q) = x.u,   sum_j a_j (x_j W_v') = (sum_j a_j x_j) W_v'
//        so the kernel only needs the raw neighbour rows x_j -- PQ codes decoded on the fly from the
//        HBM-resident store (layer 1, fused gather + decode) or the previous layer's ntgt states.
//   ('ntgt','intra','ntgt')  chain_attn  : each group of 1+l+r context nodes is a path graph with
//        self loops (token_block_dataset.py:395-398): <= 3 incoming edges per node, one wave per
//        (group, head), operands live in registers.
//   ('tgt','intra','tgt')    causal_softmax : dense causal attention over the block; the two
//        contractions run on the GEMM kernel, this is the masked row softmax in between.
#include <algorithm>
#include <cstdlib>
#include "kernels.h"

namespace gnnlm {
namespace {

constexpr int HB = 8;   // heads processed per pass of star_attn


// Sum 8 per-lane partials over the 64 lanes with a transposing butterfly: 10 shuffles instead of 48.
// On return lane 8*h holds the total of v[h].
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

// One workgroup (4 waves) per token.  QPL = float4 chunks of the feature row held per lane: lane l owns
// the contiguous floats [16*QPL*l/4 ...), i.e. (for dsub = 8, QPL = 4) the two sub-quantizers 2l, 2l+1.
// The token's kg code rows (kg x M bytes = 16 KiB at 128 x 128) are staged into LDS once with
// coalesced 16-B loads -- one full 128-B line per neighbour, the only HBM traffic of the kernel --
// so the per-neighbour dependent chain is LDS byte -> L2 centroid gather, and two neighbours are in
// flight per wave to cover the L2 latency.
//
// Measured dead ends (round 1, 2048 tokens x 128 neighbours, this kernel = 405 us): streaming the centroid
// table through LDS in 32/64-dim chunks with lane = neighbour (no cross-lane reduction) and U either on
// the scalar path (s_load -> SGPR operands; 1230 us: SGPR spills + exposed scalar latency at 1 wave/SIMD)
// or broadcast from LDS with two neighbours per lane (640 us: random 32-B LDS reads conflict, 128 barriers
// per workgroup, 1 workgroup per CU).  The L2 gather below moves a 128-B line per 32 useful bytes but keeps
// 8 waves per CU in flight; it stays until a formulation with >= 2 workgroups per CU is found.
template <int QPL>
__global__ __launch_bounds__(256) void star_attn_kernel(StarAttnParams p, bool stage_codes) {
    // stage_codes = false (k_g too large for LDS, e.g. the k_g = 1024 stress): code bytes are read from
    // the store directly
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sc = smem;                               // [HB][kg] scores, then alphas
    float* zred = smem + HB * p.kg;                 // [HB][D]  cross-wave reduction of Z
    int* okf = reinterpret_cast<int*>(zred + HB * p.D);          // [kg] neighbour validity
    uint8_t* lcodes = reinterpret_cast<uint8_t*>(okf + ((p.kg + 3) & ~3));   // [kg][M] (PQ source only)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x;
    const int D = p.D, kg = p.kg, nq = D / 4, M = p.M;
    const int64_t* ids = p.ids + (int64_t)i * kg;
    const int q_per_m = p.codes ? p.dsub / 4 : 1;

    for (int j = tid; j < kg; j += 256) okf[j] = star_nb_ok(p, i, j, ids[j]);
    __syncthreads();
    if (p.codes && stage_codes) {
        if ((M & 15) == 0) {
            const int per_row = M >> 4;
            for (int e = tid; e < kg * per_row; e += 256) {
                const int j = e / per_row, part = e - j * per_row;
                const int64_t id = ids[j];
                const int64_t lrow = p.codes_direct ? (p.codes_index ? (int64_t)p.codes_index[((int64_t)i * kg + j) * p.codes_direct] : ((int64_t)i * kg + j) * p.codes_direct) : id - p.row0;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (okf[j]) v = *reinterpret_cast<const uint4*>(p.codes + lrow * M + 16 * part);
                *reinterpret_cast<uint4*>(lcodes + j * M + 16 * part) = v;
            }
        } else {
            for (int e = tid; e < kg * M; e += 256) {
                const int j = e / M, m = e - j * M;
                const int64_t id = ids[j];
                const int64_t lrow = p.codes_direct ? (p.codes_index ? (int64_t)p.codes_index[((int64_t)i * kg + j) * p.codes_direct] : ((int64_t)i * kg + j) * p.codes_direct) : id - p.row0;
                lcodes[e] = okf[j] ? p.codes[lrow * M + m] : 0;
            }
        }
    }
    __syncthreads();

    // x_j chunk of this lane (zero for invalid neighbours / lanes beyond D)
    auto load_x = [&](int j, float4 (&x)[QPL]) {
        const bool ok = j < kg && okf[j];
#pragma unroll
        for (int t = 0; t < QPL; ++t) {
            const int q = QPL * lane + t;
            x[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok && q < nq) {
                if (p.codes) {
                    const int m = q / q_per_m;
                    const int within = (q - m * q_per_m) * 4;
                    int code;
                    if (stage_codes) {
                        code = lcodes[j * M + m];
                    } else {
                        const int64_t lrow = p.codes_direct ? (p.codes_index ? (int64_t)p.codes_index[((int64_t)i * kg + j) * p.codes_direct] : ((int64_t)i * kg + j) * p.codes_direct) : ids[j] - p.row0;
                        code = p.codes[lrow * M + m];
                    }
                    x[t] = *reinterpret_cast<const float4*>(p.centroids + ((int64_t)(m * 256 + code)) * p.dsub + within);
                } else {
                    x[t] = *reinterpret_cast<const float4*>(p.X + star_group(p, i, j) * p.x_group_stride * p.ldx + 4 * q);
                }
            }
        }
    };

    for (int h0 = 0; h0 < p.H; h0 += HB) {
        // ---- pass 1: scores s[h][j] = x_j . U[i,h,:]
        {
            float4 u[HB][QPL];
#pragma unroll
            for (int h = 0; h < HB; ++h)
#pragma unroll
                for (int t = 0; t < QPL; ++t) {
                    const int q = QPL * lane + t;
                    u[h][t] = (h0 + h < p.H && q < nq)
                                  ? *reinterpret_cast<const float4*>(p.U + ((int64_t)i * p.H + h0 + h) * D + 4 * q)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            for (int j = wave; j < kg; j += 8) {
                float4 xa[QPL], xb[QPL];
                load_x(j, xa);
                load_x(j + 4, xb);
                float pa[HB], pb[HB];
#pragma unroll
                for (int h = 0; h < HB; ++h) {
                    float a = 0.f, b = 0.f;
#pragma unroll
                    for (int t = 0; t < QPL; ++t) {
                        a = fmaf(xa[t].x, u[h][t].x, a); b = fmaf(xb[t].x, u[h][t].x, b);
                        a = fmaf(xa[t].y, u[h][t].y, a); b = fmaf(xb[t].y, u[h][t].y, b);
                        a = fmaf(xa[t].z, u[h][t].z, a); b = fmaf(xb[t].z, u[h][t].z, b);
                        a = fmaf(xa[t].w, u[h][t].w, a); b = fmaf(xb[t].w, u[h][t].w, b);
                    }
                    pa[h] = a;
                    pb[h] = b;
                }
                const float ta = reduce8(pa, lane), tb = reduce8(pb, lane);
                if ((lane & 7) == 0) {
                    sc[(lane >> 3) * kg + j] = okf[j] ? ta : -INFINITY;
                    if (j + 4 < kg) sc[(lane >> 3) * kg + j + 4] = okf[j + 4] ? tb : -INFINITY;
                }
            }
        }
        __syncthreads();
        // ---- softmax over j per head (wave w: heads w, w+4)
        for (int h = wave; h < HB; h += 4) {
            float mx = -INFINITY;
            for (int j = lane; j < kg; j += 64) mx = fmaxf(mx, sc[h * kg + j]);
            mx = wave_max(mx);
            float sum = 0.f;
            int cnt = 0;
            for (int j = lane; j < kg; j += 64) {
                const float s = sc[h * kg + j];
                const bool ok = s != -INFINITY;
                const float e = ok ? expf(s - mx) : 0.f;
                cnt += ok;
                sum += e;
                sc[h * kg + j] = e;
            }
            sum = wave_sum(sum);
            const float inv = sum > 0.f ? 1.f / sum : 0.f;
            for (int j = lane; j < kg; j += 64) sc[h * kg + j] *= inv;
            if (h == 0 && h0 == 0) {
                cnt = (int)wave_sum((float)cnt);
                if (lane == 0 && p.has_nb) p.has_nb[i] = cnt > 0 ? 1.f : 0.f;
            }
        }
        __syncthreads();
        // ---- pass 2: Z[h,:] = sum_j alpha[h][j] x_j   (alpha = 0 and x = 0 for invalid neighbours)
        {
            float4 z[HB][QPL];
#pragma unroll
            for (int h = 0; h < HB; ++h)
#pragma unroll
                for (int t = 0; t < QPL; ++t) z[h][t] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int j = wave; j < kg; j += 8) {
                float4 xa[QPL], xb[QPL];
                load_x(j, xa);
                load_x(j + 4, xb);
                const bool hb = j + 4 < kg;
#pragma unroll
                for (int h = 0; h < HB; ++h) {
                    const float a = sc[h * kg + j];
                    const float b = hb ? sc[h * kg + j + 4] : 0.f;
#pragma unroll
                    for (int t = 0; t < QPL; ++t) {
                        z[h][t].x = fmaf(a, xa[t].x, z[h][t].x); z[h][t].x = fmaf(b, xb[t].x, z[h][t].x);
                        z[h][t].y = fmaf(a, xa[t].y, z[h][t].y); z[h][t].y = fmaf(b, xb[t].y, z[h][t].y);
                        z[h][t].z = fmaf(a, xa[t].z, z[h][t].z); z[h][t].z = fmaf(b, xb[t].z, z[h][t].z);
                        z[h][t].w = fmaf(a, xa[t].w, z[h][t].w); z[h][t].w = fmaf(b, xb[t].w, z[h][t].w);
                    }
                }
            }
            // deterministic cross-wave reduction: waves add in order 0,1,2,3
            for (int w = 0; w < 4; ++w) {
                if (wave == w) {
#pragma unroll
                    for (int h = 0; h < HB; ++h)
#pragma unroll
                        for (int t = 0; t < QPL; ++t) {
                            const int q = QPL * lane + t;
                            if (q < nq) {
                                float4* dst = reinterpret_cast<float4*>(zred + h * D + 4 * q);
                                if (w == 0) {
                                    *dst = z[h][t];
                                } else {
                                    float4 c = *dst;
                                    c.x += z[h][t].x; c.y += z[h][t].y; c.z += z[h][t].z; c.w += z[h][t].w;
                                    *dst = c;
                                }
                            }
                        }
                }
                __syncthreads();
            }
            for (int e = tid; e < HB * nq; e += 256) {
                const int h = e / nq, q = e - h * nq;
                if (h0 + h < p.H)
                    *reinterpret_cast<float4*>(p.Z + ((int64_t)i * p.H + h0 + h) * D + 4 * q) =
                        *reinterpret_cast<const float4*>(zred + h * D + 4 * q);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// Chunk-swept star attention for the PQ source (layer 1): k_g <= 128, dsub in {4, 8}, D % 32 == 0,
// <= 8 heads per launch, one token per workgroup of 4 waves.
//
// The generic kernel above lets every lane own a fixed slice of the 1-MiB centroid table, so a wave's
// gathers spray over the whole table: every 32-B row costs a 128-B line from L2 (measured 1.57 ms per
// 8192 tokens; 50 % of the wave time is spent in s_waitcnt, TA stalled by the L2).  Here the four waves of
// a workgroup sweep the feature dimension in 32-dim chunks together.  Per chunk the token's 128 x 32 slab
// of decoded neighbour features is built ONCE in LDS by all 256 threads (4 independent 16-B gathers per
// thread, all inside the chunk's 4 sub-tables = 32 KiB, which stay in the CU's L1); the math only reads LDS.
// Everything the loop waits for is two chunks ahead of its use: the gathers of chunk c+2 and the query
// slice U of chunk c+1 are in flight (in a second register set; the loop is unrolled by two so the sets
// have static names) while chunk c is computed, and the slab of chunk c+1 is committed to the other LDS
// buffer before the chunk's single barrier.
//   pass 1 (f32 matrix cores): S[128 nb x 16 (8 real) heads] += X[128 x 32] . U^T per chunk as
//           v_mfma_f32_16x16x4_f32; wave w owns neighbours 32w..32w+31 (two accumulator tiles, 16 MFMAs per
//           chunk); the chunk's query rows U[h] ride in the slab as rows 128..135.  The scores come out of
//           the accumulators whole: no cross-lane reduction (the VALU formulation spent 64 v_pk_fma + ~100
//           VALU bookkeeping instructions per chunk per wave and 96 DPP shuffles per token here).
//   softmax per head over the neighbours (LDS).
//   pass 2 (f32 matrix cores): per chunk Z[16 (8 real) heads x 32 dims] = alpha^T [16 x 128] . X [128 x 32]
//           as v_mfma_f32_16x16x4_f32.  Wave w owns column tile w & 1 and the neighbour half w >> 1
//           (16 MFMAs per chunk); the A operand (alpha) lives in 16 registers for the whole sweep, the B
//           operand is ONE ds_read_b32 per MFMA, so the slab is read exactly once (a VALU formulation with
//           lanes = (dq, head, neighbour group) re-reads it once per head: 128 KiB of LDS reads per chunk).
//           The two neighbour halves meet through a 2-KiB LDS buffer after the chunk's barrier.
// Slab rows are 36 floats apart: 36 r mod 64 runs over the 16 multiples of 4, so the pass-1 operand read
// (16 rows x 4 consecutive dims) touches 64 different banks; pass 2 groups the neighbours 4 apart into one
// k-step (rows 4 apart = 16 banks apart: 4 rows x 16 consecutive dims, again 64 different banks).
// Measured at T = 8192, WikiText-103 shape: generic 1.57 ms; swept, VALU pass 2, depth-1 prefetch 1.42 ms;
// + MFMA pass 2 1.36 ms; + depth-2 prefetch 1.39 ms (no gain: not latency-bound); + MFMA pass 1 1.17-1.19 ms.
// + producer / consumer wave roles 1.16 ms: the iteration is as long as the decode chain alone -- the kernel is bound
// by the rate at which the L1 serves 16-B gathers (~0.37 lines per cycle per CU measured here), not by latency.
// A timing-only variant without the 17-KiB code staging (3 workgroups per CU instead of 2) ran 1.01 ms, but
// feeding it needs the codes transposed per (token, chunk) by a pre-pass that moves 2 x 134 MB -- no net gain.
// Mapping one gather instruction to ONE sub-table (32 neighbours x the two halves of a row, so that rows sharing
// a 128-B line coalesce: ~25 instead of 32 lines per instruction) measured 1.19-1.22 ms against 1.17: the look-up
// rate does not improve with intra-instruction line sharing.
#ifndef GNNLM_STAR_EXP
#define GNNLM_STAR_EXP 0
#endif
template <int DSUB>
__global__ __launch_bounds__(512, 2) void star_attn_sweep_kernel(StarAttnParams p, int h0) {
    // Wave specialisation: waves 0..3 are CONSUMERS (the two MFMA passes and the softmax), waves 4..7 PRODUCERS
    // (the decode: code look-up, centroid gathers, slab commits).  The two halves meet only at the chunk barrier,
    // so an iteration costs max(decode chain, MFMA chain) instead of their sum.  With every wave running both
    // chains back to back, waves spent ~45 % of their time in s_waitcnt (SQ_WAIT_ANY) while neither the L1 tag
    // pipeline (~37 % busy) nor the matrix cores (~40 %) were saturated: 1.17 ms per 8192 tokens.  The roles are
    // separate code paths (same number of barriers on both), so a wave's registers hold one role's state only.
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int KGM = 128, CD = 32;
    constexpr int XS = CD + 4;                  // slab row stride (floats), see the bank notes above
    constexpr int SLAB = (KGM + HB) * XS;       // 128 neighbour rows + the 8 query rows U[h] of the chunk
    constexpr int NDQ = CD / 4;                 // float4 per chunk row
    constexpr int SCS = KGM + 4;                // score row stride: heads land on different banks
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int D = p.D, NCH = D / CD;
    const int kg = p.kg, M = p.M, H = p.H;
    const int MS = M + 4;                       // padded code row stride (bytes): conflict-free column reads
    const bool producer = threadIdx.x >= 256;
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;      // index inside the role
    float* xc = smem;                                               // [2][KGM + HB][XS] decoded slabs + query rows
    float* sc = xc + 2 * SLAB;                                      // [HB][SCS] scores -> alphas
    uint8_t* lcodes = reinterpret_cast<uint8_t*>(sc + HB * SCS);    // [KGM][MS]
    float* zpart = reinterpret_cast<float*>(lcodes + ((KGM * MS + 15) & ~15));     // [2][2 tiles][4][64]
    const int i = blockIdx.x;
    const int64_t* ids = p.ids + (int64_t)i * kg;

    {   // stage the code rows (zeros for invalid neighbours): 16-B global pieces, coalesced; all 512 threads
        const int per_row = M >> 4;
        for (int e = threadIdx.x; e < KGM * per_row; e += 512) {
            const int j = e / per_row, part = e - j * per_row;
            const int64_t id = j < kg ? ids[j] : -1;
            const int64_t lrow = p.codes_direct ? (p.codes_index ? (int64_t)p.codes_index[((int64_t)i * kg + j) * p.codes_direct] : ((int64_t)i * kg + j) * p.codes_direct) : id - p.row0;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (j < kg && star_nb_ok(p, i, j, id)) v = *reinterpret_cast<const uint4*>(p.codes + lrow * M + 16 * part);
            uint32_t* dst = reinterpret_cast<uint32_t*>(lcodes + j * MS + 16 * part);
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
    }
    __syncthreads();

    if (producer) {
        // ================================================================ PRODUCERS: build the slabs
        // thread t decodes float4 `t % 8` of the neighbours (t / 8) + 32 q, q = 0..3; lane pairs fetch the two halves
        // of one 32-B centroid row.  Threads 0..63 also carry the chunk's query rows (pass 1): float4 `t % 8` of
        // head t / 8.  Gathers are two chunks ahead of their commit (two register sets, loop unrolled by two).
        const float* cen = p.centroids;
        const int ddq = tid % NDQ, dj0 = tid / NDQ;
        const int dm_in = (4 * ddq) / DSUB, dwithin = (4 * ddq) % DSUB;
        const float* Uq = p.U + ((int64_t)i * H + h0 + min(dj0 & 7, H - 1 - h0)) * D + 4 * ddq;
        float4 xrA0, xrA1, xrA2, xrA3, xrB0, xrB1, xrB2, xrB3, uqA, uqB;
        uqA = uqB = make_float4(0.f, 0.f, 0.f, 0.f);
#define GNNLM_FETCH1(XR, Q, m_)                                                                         \
    XR = *reinterpret_cast<const float4*>(                                                              \
        cen + ((int64_t)((m_) * 256 + (GNNLM_STAR_EXP == 3 ? 0 : lcodes[(dj0 + 32 * (Q)) * MS + (m_)]))) * DSUB + dwithin);
#define GNNLM_FETCH(c, S, WITH_U)                                                                       \
    if ((c) < NCH) {                                                                                    \
        const int m_ = (c) * (CD / DSUB) + dm_in;                                                       \
        GNNLM_FETCH1(xr##S##0, 0, m_) GNNLM_FETCH1(xr##S##1, 1, m_)                                     \
        GNNLM_FETCH1(xr##S##2, 2, m_) GNNLM_FETCH1(xr##S##3, 3, m_)                                     \
        if (WITH_U && tid < 64) uq##S = *reinterpret_cast<const float4*>(Uq + (c) * CD);                \
    }
#define GNNLM_COMMIT(buf, S, c, WITH_U)                                                                 \
    if ((c) < NCH) {                                                                                    \
        float* d_ = xc + (buf) * SLAB + dj0 * XS + 4 * ddq;                                             \
        *reinterpret_cast<float4*>(d_) = xr##S##0;                                                      \
        *reinterpret_cast<float4*>(d_ + 32 * XS) = xr##S##1;                                            \
        *reinterpret_cast<float4*>(d_ + 64 * XS) = xr##S##2;                                            \
        *reinterpret_cast<float4*>(d_ + 96 * XS) = xr##S##3;                                            \
        if (WITH_U && tid < 64) *reinterpret_cast<float4*>(d_ + 128 * XS) = uq##S;                      \
    }
#define GNNLM_PRODUCE_SWEEP(WITH_U)                                                                     \
    GNNLM_FETCH(0, A, WITH_U)                                                                           \
    GNNLM_COMMIT(0, A, 0, WITH_U)                                                                       \
    GNNLM_FETCH(1, B, WITH_U)                                                                           \
    __syncthreads();                                                                                    \
    for (int c = 0; c < NCH; c += 2) {                                                                  \
        GNNLM_FETCH(c + 2, A, WITH_U)                                                                   \
        GNNLM_COMMIT(1, B, c + 1, WITH_U)                                                               \
        __syncthreads();                                                                                \
        if (c + 1 < NCH) {                                                                              \
            GNNLM_FETCH(c + 3, B, WITH_U)                                                               \
            GNNLM_COMMIT(0, A, c + 2, WITH_U)                                                           \
            __syncthreads();                                                                            \
        }                                                                                               \
    }
        GNNLM_PRODUCE_SWEEP(true)               // pass 1 (query rows ride along)
        __syncthreads();                        // scores written (consumers)
        GNNLM_PRODUCE_SWEEP(false)              // pass 2: its first slabs are built under the consumers' softmax
#undef GNNLM_PRODUCE_SWEEP
#undef GNNLM_FETCH
#undef GNNLM_FETCH1
#undef GNNLM_COMMIT
        return;
    }

    // ==================================================================== CONSUMERS: the two MFMA passes
    const int n16 = lane & 15, g = lane >> 4;
    // ---------------- pass 1: S[128 nb x 16 (8 real) heads] += X[128 x 32] . U^T
    {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const float* xa = xc + (32 * wave + n16) * XS + g;          // A: X[nb = 32 w + 16 rt + n16][4 ks + g]
        const float* ub = xc + (KGM + (n16 & 7)) * XS + g;          // B: U[head n16][4 ks + g] (columns 8..15: unused copies)
#define GNNLM_PASS1(buf)                                                                                \
    _Pragma("unroll") for (int ks = 0; ks < (GNNLM_STAR_EXP == 1 ? 0 : 8); ++ks) {                      \
        const float b_ = ub[(buf) * SLAB + 4 * ks];                                                     \
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[(buf) * SLAB + 4 * ks], b_, acc0, 0, 0, 0);      \
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[(buf) * SLAB + 16 * XS + 4 * ks], b_, acc1, 0, 0, 0); \
    }
        __syncthreads();                        // slab 0 built
        for (int c = 0; c < NCH; c += 2) {
            GNNLM_PASS1(0)
            __syncthreads();
            if (c + 1 < NCH) {
                GNNLM_PASS1(1)
                __syncthreads();
            }
        }
#undef GNNLM_PASS1
        // C layout: acc_rt[r] = S[nb = 32 w + 16 rt + 4 g + r][head n16]
        if (n16 < HB) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j0 = 32 * wave + 4 * g + r, j1 = j0 + 16;
                sc[n16 * SCS + j0] = (j0 < kg && star_nb_ok(p, i, j0, ids[j0])) ? acc0[r] : -INFINITY;
                sc[n16 * SCS + j1] = (j1 < kg && star_nb_ok(p, i, j1, ids[j1])) ? acc1[r] : -INFINITY;
            }
        }
    }
    __syncthreads();
    // ---------------- softmax over j per head (wave w: heads w, w+4) while the producers build pass 2's first slab
    for (int h = wave; h < HB; h += 4) {
        const float v0 = sc[h * SCS + lane], v1 = sc[h * SCS + 64 + lane];
        const float mx = wave_max(fmaxf(v0, v1));
        const float e0 = v0 == -INFINITY ? 0.f : expf(v0 - mx), e1 = v1 == -INFINITY ? 0.f : expf(v1 - mx);
        const float sum = wave_sum(e0 + e1);
        const float inv = sum > 0.f ? 1.f / sum : 0.f;
        sc[h * SCS + lane] = e0 * inv;
        sc[h * SCS + 64 + lane] = e1 * inv;
        if (h == 0 && h0 == 0 && lane == 0 && p.has_nb) p.has_nb[i] = sum > 0.f ? 1.f : 0.f;
    }
    // ---------------- pass 2: Z[16 (8 real) heads x 32 dims] = alpha^T . X per chunk
    {
        const int tile = wave & 1, kh = wave >> 1;
        __syncthreads();                       // slab 0 of pass 2 built; also orders the softmax writes before the alpha reads
        // k-step ks, lane group g  <->  neighbour j = 64 kh + 16 (ks / 4) + (ks % 4) + 4 g: rows 4 apart are
        // 16 banks apart at a 36-float stride
        float a_reg[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
            a_reg[ks] = n16 < HB ? sc[n16 * SCS + 64 * kh + 16 * (ks >> 2) + (ks & 3) + 4 * g] : 0.f;
        const float* xb0 = xc + (64 * kh + 4 * g) * XS + 16 * tile + n16;
        float* zp0 = zpart + tile * 256 + lane;
        float* zo0 = p.Z + ((int64_t)i * H + h0 + 4 * g) * D + 16 * tile + n16;
        // C layout of the 16x16 tile: acc[r] = Z[head 4g + r][dim 16 tile + n16]
#define GNNLM_PASS2(buf, c)                                                                             \
    {                                                                                                   \
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};                                                               \
        _Pragma("unroll") for (int ks = 0; ks < (GNNLM_STAR_EXP == 2 ? 0 : 16); ++ks)                   \
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_reg[ks], xb0[(buf) * SLAB + (16 * (ks >> 2) + (ks & 3)) * XS], acc, 0, 0, 0); \
        float* zp = zp0 + (buf) * 512;                                                                  \
        if (kh == 1) { zp[0] = acc[0]; zp[64] = acc[1]; zp[128] = acc[2]; zp[192] = acc[3]; }           \
        zacc = acc;                                                                                     \
    }
#define GNNLM_ZSTORE(buf, c)                                                                            \
    if (kh == 0 && g < 2) {                                                                             \
        const float* zp = zp0 + (buf) * 512;                                                            \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                   \
            if (h0 + 4 * g + r < H) zo0[(int64_t)r * D + (c) * CD] = zacc[r] + zp[64 * r];              \
    }
        f32x4 zacc;
        for (int c = 0; c < NCH; c += 2) {
            GNNLM_PASS2(0, c)
            __syncthreads();
            GNNLM_ZSTORE(0, c)
            if (c + 1 < NCH) {
                GNNLM_PASS2(1, c + 1)
                __syncthreads();
                GNNLM_ZSTORE(1, c + 1)
            }
        }
#undef GNNLM_PASS2
#undef GNNLM_ZSTORE
    }
}

constexpr int MAX_NG = 8;
constexpr int MAX_EPT = 4;

// One wave per (group, head).  Position order along the path: o-l .. o-1, o, o+1 .. o+r.
__global__ __launch_bounds__(256) void chain_attn_kernel(ChainAttnParams p) {
    const int lane = threadIdx.x & 63;
    // (a device-side group count -- ABI 9 -- comes with a capped grid: the waves then walk the tasks)
    const int64_t n_tasks = (p.n_groups_dev ? min(p.n_groups, (int64_t)*p.n_groups_dev) : p.n_groups) * p.H;
  for (int64_t task = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); task < n_tasks; task += (int64_t)gridDim.x * 4) {
    const int64_t g = task / p.H;
    const int h = (int)(task - g * p.H);
    const int n_g = 1 + p.left + p.right;
    const int dk = p.dk;
    const float scale = p.scale ? p.scale[h] : 1.f;

    // radius_p1 = r + 1 > 0: only the positions within r of the centre are destinations (a later layer reads nothing else);
    // their sources lie within r + 1, and only those rows of Q / K / V exist
    const int rad = p.radius_p1 > 0 ? p.radius_p1 - 1 : MAX_NG;
    float q[MAX_NG][MAX_EPT], k[MAX_NG][MAX_EPT], v[MAX_NG][MAX_EPT];
    bool ok[MAX_NG], dst[MAX_NG];
    int64_t slot_of[MAX_NG];
#pragma unroll
    for (int pos = 0; pos < MAX_NG; ++pos) {
        ok[pos] = dst[pos] = false;
        slot_of[pos] = 0;
        if (pos < n_g) {
            const int c = pos < p.left ? pos + 1 : (pos == p.left ? 0 : pos);
            const int64_t s = g * n_g + c;
            const int dist = pos < p.left ? p.left - pos : pos - p.left;
            slot_of[pos] = s;
            dst[pos] = dist <= rad;
            // (ABI 11: K / V keyed by datastore row -- slot s reads row kv_index[s]; a valid slot always has one.  The index is loaded
            // BESIDE the validity byte, not behind it: gated by `ok` it was a third dependent round trip per task and the layer-0 launch of
            // the 3-layer recipe took 20 ms instead of 10)
            const int32_t kvr = (p.kv_index && dist <= rad + 1) ? p.kv_index[s] : 0;
            ok[pos] = dist <= rad + 1 && p.valid[s] != 0;
            const int64_t row_kv = p.kv_index ? (int64_t)max(kvr, 0) : s;
#pragma unroll
            for (int t = 0; t < MAX_EPT; ++t) {
                const int e = lane + 64 * t;
                const bool in = ok[pos] && e < dk;
                const int64_t off = s * p.ld + h * dk + (in ? e : 0);
                const int64_t off_kv = row_kv * p.ld + h * dk + (in ? e : 0);
                q[pos][t] = in && dst[pos] ? p.Q[off] : 0.f;
                k[pos][t] = in ? p.K[off_kv] : 0.f;
                v[pos][t] = in ? p.V[off_kv] : 0.f;
            }
        }
    }
#pragma unroll
    for (int pos = 0; pos < MAX_NG; ++pos) {
        if (pos >= n_g) break;
        if (!dst[pos]) continue;
        float sc[3];
        bool has[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int u = pos + d - 1;
            has[d] = false;
            sc[d] = -INFINITY;
            if (u >= 0 && u < MAX_NG) {
                if (u < n_g && ok[u] && ok[pos]) {
                    float a = 0.f;
#pragma unroll
                    for (int t = 0; t < MAX_EPT; ++t) a = fmaf(q[pos][t], k[u][t], a);
                    a = wave_sum(a);
                    sc[d] = a * scale;
                    has[d] = true;
                }
            }
        }
        const float mx = fmaxf(sc[0], fmaxf(sc[1], sc[2]));
        float e[3], den = 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            e[d] = has[d] ? expf(sc[d] - mx) : 0.f;
            den += e[d];
        }
        const float inv = den > 0.f ? 1.f / den : 0.f;
#pragma unroll
        for (int t = 0; t < MAX_EPT; ++t) {
            const int el = lane + 64 * t;
            if (el < dk) {
                float o = 0.f;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const int u = pos + d - 1;
                    if (u >= 0 && u < MAX_NG) {
                        if (has[d]) o = fmaf(e[d] * inv, v[u][t], o);
                    }
                }
                p.out[slot_of[pos] * p.ldo + h * dk + el] = o;
            }
        }
    }
  }
}

// One wave per score row.  Row w of matrix m: keep u <= w (and w-u < max_ctx), softmax, zero the rest
// (including the padding columns up to ld, which the following P.V GEMM reads).
__global__ __launch_bounds__(256) void causal_softmax_kernel(float* S, int64_t n_rows, int T, int64_t ld, int max_ctx) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int w = (int)(row % T);
    float* r = S + row * ld;          // matrices are stored with T rows each, contiguous
    const int lo = max_ctx > 0 ? max(0, w - max_ctx + 1) : 0;
    float mx = -INFINITY;
    for (int u = lo + lane; u <= w; u += 64) mx = fmaxf(mx, r[u]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int u = lo + lane; u <= w; u += 64) sum += expf(r[u] - mx);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int u = lane; u < (int)ld; u += 64) r[u] = (u >= lo && u <= w) ? expf(r[u] - mx) * inv : 0.f;
}


// ---------------------------------------------------------------------------------------------------
// Fused causal ('tgt','intra','tgt') attention for the recipe shape T = 256, d_k = 128: one workgroup of 8
// waves per (block, head); scores, masked softmax and P.V without the [T, T] score matrix ever leaving the
// registers (the GEMM + softmax + GEMM formulation writes and re-reads 67 MB of scores per 8192 tokens).
//
// Wave w owns 32 queries.  Pass 1 computes S^T = K' Q^T with v_mfma_f32_32x32x2_f32 -- accumulator rows are
// KEYS, columns (lane & 31) are queries -- so a query's 256 scores are registers of two lanes and the masked
// softmax is in-lane plus one v_permlane32_swap.  In that layout accumulator register r of key tile t in lane
// (query, half) is exactly the A operand of the 32x32x2 MFMA whose k pair is (key(t,r,0), key(t,r,1)): pass 2
// feeds the probabilities to P.V straight from the accumulators, B = two rows of V from LDS.  K' and V stream
// through LDS in blocks of 64 keys (double-buffered); key tiles above the diagonal are skipped, and the
// query tiles are dealt to the waves so that the two waves of a SIMD carry equal work (tiles w and 7 - w).
// The score scale and the relation prior are folded into K' by prepare_hgt_weights, as in the GEMM path.
typedef float f32x16 __attribute__((ext_vector_type(16)));
struct CausalAttnParams {
    const float* Q; const float* K; const float* V; float* out;    // [n_blocks * T, ld] rows; head h at column h * dk
    int64_t ld, ldo;
    int n_blocks, H, max_ctx;
    int accumulate;               // 1: out += result (the cross-edge-type sum of the HGT layer lands in one buffer)
};

__global__ __launch_bounds__(512, 1) void causal_attn_256x128_kernel(CausalAttnParams p) {
    constexpr int T = 256, DK = 128, KB = 64;
    constexpr int KS = DK + 4;                    // K' block row stride: conflict-free ds_read_b128 of 16 rows
    constexpr int VS = DK + 8;                    // V block row stride: rows 4 apart are 32 banks apart
    constexpr int BUF = KB * VS;
    extern __shared__ __attribute__((aligned(16))) float smem[];    // [2][KB][VS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l32 = lane & 31;
    const int blk = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int qt = wave < 4 ? wave : 11 - wave;   // waves w and w + 4 share a SIMD: tiles (0,7) (1,6) (2,5) (3,4)
    const int q0 = 32 * qt, query = q0 + l32;
    const int64_t row0 = (int64_t)blk * T;
    const float* Qb = p.Q + row0 * p.ld + h * DK;
    const float* Kb = p.K + row0 * p.ld + h * DK;
    const float* Vb = p.V + row0 * p.ld + h * DK;

    // B operand of pass 1: Q[query][8 s + 4 half + e], e = 0..3, s = 0..15 (the k permutation of the staged K')
    float4 qreg[16];
#pragma unroll
    for (int s_ = 0; s_ < 16; ++s_)
        qreg[s_] = *reinterpret_cast<const float4*>(Qb + (int64_t)query * p.ld + 8 * s_ + 4 * half);

    f32x16 sc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[t][r] = 0.f;

    // staging: 512 threads x 4 float4 = one 64 x 128 block
    const int srow = tid >> 5, scol = (tid & 31) * 4;           // rows srow + 16 u, u = 0..3
    float4 stg[4];
#define GNNLM_CA_LOAD(src, kb)                                                                          \
    _Pragma("unroll") for (int u = 0; u < 4; ++u)                                                       \
        stg[u] = *reinterpret_cast<const float4*>((src) + (int64_t)((kb) * KB + srow + 16 * u) * p.ld + scol);
#define GNNLM_CA_STORE(buf, stride)                                                                     \
    _Pragma("unroll") for (int u = 0; u < 4; ++u)                                                       \
        *reinterpret_cast<float4*>(smem + (buf) * BUF + (srow + 16 * u) * (stride) + scol) = stg[u];

    // ---------------- pass 1: S^T tiles (keys x queries)
    GNNLM_CA_LOAD(Kb, 0)
    GNNLM_CA_STORE(0, KS)
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < T / KB; ++kb) {
        if (kb + 1 < T / KB) GNNLM_CA_LOAD(Kb, kb + 1)
        const float* kl = smem + (kb & 1) * BUF;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int t = 2 * kb + tt;
            if (t <= qt) {
                const float* kr = kl + (32 * tt + l32) * KS + 4 * half;
#pragma unroll
                for (int s_ = 0; s_ < 16; ++s_) {
                    const float4 a = *reinterpret_cast<const float4*>(kr + 8 * s_);
                    sc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qreg[s_].x, sc[t], 0, 0, 0);
                    sc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qreg[s_].y, sc[t], 0, 0, 0);
                    sc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qreg[s_].z, sc[t], 0, 0, 0);
                    sc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qreg[s_].w, sc[t], 0, 0, 0);
                }
            }
        }
        if (kb + 1 < T / KB) GNNLM_CA_STORE((kb + 1) & 1, KS)
        __syncthreads();
    }
    // ---------------- masked softmax over the keys of each query (edge_softmax by destination, hgt.py:356)
    GNNLM_CA_LOAD(Vb, 0)                              // V block 0 flies under the softmax
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
            const bool ok = t <= qt && key <= query && (p.max_ctx <= 0 || query - key < p.max_ctx);
            sc[t][r] = ok ? sc[t][r] : -INFINITY;
            mx = fmaxf(mx, sc[t][r]);
        }
    {
        const gnnlm_u32x2 x = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(__uint_as_float(x.x), __uint_as_float(x.y));
    }
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = sc[t][r] == -INFINITY ? 0.f : expf(sc[t][r] - mx);
            sc[t][r] = e;
            sum += e;
        }
    {
        const gnnlm_u32x2 x = __builtin_amdgcn_permlane32_swap(__float_as_uint(sum), __float_as_uint(sum), false, false);
        sum = __uint_as_float(x.x) + __uint_as_float(x.y);
    }
    const float inv = 1.f / sum;                      // the diagonal key is always valid: sum >= 1
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[t][r] *= inv;

    // ---------------- pass 2: out[query][:] = sum_key P[query][key] V[key][:]
    f32x16 oc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) oc[c][r] = 0.f;
    GNNLM_CA_STORE(0, VS)
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < T / KB; ++kb) {
        if (kb + 1 < T / KB) GNNLM_CA_LOAD(Vb, kb + 1)
        const float* vl = smem + (kb & 1) * BUF + l32;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int t = 2 * kb + tt;
            if (t <= qt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float* vr = vl + (32 * tt + (r & 3) + 8 * (r >> 2) + 4 * half) * VS;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        oc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(sc[t][r], vr[32 * c], oc[c], 0, 0, 0);
                }
            }
        }
        if (kb + 1 < T / KB) GNNLM_CA_STORE((kb + 1) & 1, VS)
        __syncthreads();
    }
#undef GNNLM_CA_LOAD
#undef GNNLM_CA_STORE
    float* ob = p.out + (row0 + q0) * p.ldo + h * DK + l32;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (p.accumulate) ob[(int64_t)((r & 3) + 8 * (r >> 2) + 4 * half) * p.ldo + 32 * c] += oc[c][r];
            else ob[(int64_t)((r & 3) + 8 * (r >> 2) + 4 * half) * p.ldo + 32 * c] = oc[c][r];
}

}  // namespace

int star_attn(const StarAttnParams& p, hipStream_t stream) {
    GNNLM_REQUIRE(p.U && p.ids && p.Z, "star_attn: null operand");
    GNNLM_REQUIRE(p.T >= 0 && p.H > 0 && p.kg > 0 && p.D > 0 && p.D % 4 == 0 && p.D <= 1024,
                  "star_attn: need D % 4 == 0 and D <= 1024");
    const bool pq = p.codes != nullptr || p.shards != nullptr;
    GNNLM_REQUIRE(pq != (p.X != nullptr), "star_attn: exactly one of codes (or mapped shards) / X");
    GNNLM_REQUIRE(!p.shards || (!p.codes_direct && p.n_store > 0), "star_attn: a shard table needs n_store and no fetched codes");
    if (pq) {
        GNNLM_REQUIRE(p.centroids && p.M > 0 && p.dsub % 4 == 0 && p.M * p.dsub == p.D,
                      "star_attn: PQ source needs centroids and M*dsub == D, dsub % 4 == 0");
    } else {
        GNNLM_REQUIRE(p.ldx % 4 == 0 && (uintptr_t)p.X % 16 == 0, "star_attn: X alignment");
    }
    if (p.T == 0) return OK;
    const size_t base_lds = (size_t)(HB * p.kg + HB * p.D + ((p.kg + 3) & ~3)) * sizeof(float);
    const size_t code_lds = p.codes ? (((size_t)p.kg * p.M + 15) & ~size_t(15)) : 0;
    const bool stage = base_lds + code_lds <= 64 * 1024;          // keep >= 2 workgroups per CU
    const size_t shmem = base_lds + (stage ? code_lds : 0);
    GNNLM_REQUIRE(shmem <= 160 * 1024, "star_attn: kg too large for LDS");
    const int nq = p.D / 4;
    const double rows = (double)p.T * p.kg;
    GNNLM_REQUIRE(!p.shards || star_attn_tab_eligible(p), "star_attn: mapped shards need the table-resident kernel (k_g <= 128, dsub 4 / 8, M % 16 == 0)");
    if (star_attn_tab_eligible(p) && (p.shards || (!getenv("GNNLM_STAR_SWEEP") && !getenv("GNNLM_STAR_GENERIC")))) {
        // table-resident formulation (star_tab.hip): the default for the PQ source with k_g <= 128
        ProfScope prof(K_STAR, stream, 4.0 * rows * p.H * p.D, rows * (8.0 + p.M) + 8.0 * p.T * p.H * p.D);
        return star_attn_tab(p, stream);
    }
    if (p.codes && p.kg <= 128 && (p.dsub == 4 || p.dsub == 8) && p.M % 16 == 0 && p.D % 32 == 0 &&
        (uintptr_t)p.codes % 16 == 0 && !getenv("GNNLM_STAR_GENERIC")) {
        ProfScope prof(K_STAR, stream, 4.0 * rows * p.H * p.D, rows * (8.0 + p.M) + 8.0 * p.T * p.H * p.D);
        const int MS = p.M + 4;
        const size_t lds_bytes = 4 * (size_t)(2 * (128 + HB) * 36 + HB * 132 + (128 * MS + 15) / 4 + 1024);
        for (int h0 = 0; h0 < p.H; h0 += HB) {
            dim3 grid(p.T), block(512);
            if (p.dsub == 8) hipLaunchKernelGGL((star_attn_sweep_kernel<8>), grid, block, lds_bytes, stream, p, h0);
            else hipLaunchKernelGGL((star_attn_sweep_kernel<4>), grid, block, lds_bytes, stream, p, h0);
        }
        GNNLM_LAUNCH_CHECK();
        return OK;
    }
    ProfScope prof(K_STAR, stream, 4.0 * rows * p.H * p.D,
                   rows * (8.0 + (p.codes ? (double)p.M : 4.0 * p.D)) + 8.0 * p.T * p.H * p.D);
    if (star_attn_dense_eligible(p)) return star_attn_dense(p, stream);      // dense rows at the recipe's widths: one pass (star_dense.hip)
    dim3 grid(p.T), block(256);
    if (nq <= 64) {
        hipLaunchKernelGGL(star_attn_kernel<1>, grid, block, shmem, stream, p, stage);
    } else if (nq <= 128) {
        hipLaunchKernelGGL(star_attn_kernel<2>, grid, block, shmem, stream, p, stage);
    } else {
        hipLaunchKernelGGL(star_attn_kernel<4>, grid, block, shmem, stream, p, stage);
    }
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int chain_attn(const ChainAttnParams& p, hipStream_t stream) {
    GNNLM_REQUIRE(p.Q && p.K && p.V && p.valid && p.out, "chain_attn: null operand");
    GNNLM_REQUIRE(1 + p.left + p.right <= MAX_NG, "chain_attn: 1+left+right must be <= 8");
    GNNLM_REQUIRE(p.dk > 0 && p.dk <= 64 * MAX_EPT && p.H > 0, "chain_attn: d_k must be <= 256");
    const int64_t tasks = p.n_groups * p.H;
    if (tasks == 0) return OK;
    const double slots = (double)p.n_groups * (1 + p.left + p.right);
    ProfScope prof(K_CHAIN, stream, slots * p.H * p.dk * 12.0, slots * (16.0 * p.H * p.dk + 1.0));
    const int64_t wgs = cdiv(tasks, 4);
    hipLaunchKernelGGL(chain_attn_kernel, dim3((unsigned)(p.n_groups_dev ? std::min<int64_t>(wgs, 256 * 32) : wgs)), dim3(256), 0, stream, p);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int causal_softmax(float* S, int64_t n_mats, int T, int64_t ld, int max_ctx, hipStream_t stream) {
    GNNLM_REQUIRE(S && T > 0 && ld >= T, "causal_softmax: bad arguments");
    const int64_t rows = n_mats * T;
    if (rows == 0) return OK;
    ProfScope prof(K_CAUSAL, stream, 0.0, 8.0 * rows * ld);
    hipLaunchKernelGGL(causal_softmax_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, stream, S, rows, T, ld, max_ctx);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

bool causal_attn_fused_ok(int T, int dk) { return T == 256 && dk == 128; }

int causal_attn_fused(const float* Q, const float* K, const float* V, int64_t ld, float* out, int64_t ldo,
                      int n_blocks, int T, int H, int dk, int max_ctx, hipStream_t stream, bool accumulate) {
    GNNLM_REQUIRE(Q && K && V && out, "causal_attn: null operand");
    GNNLM_REQUIRE(causal_attn_fused_ok(T, dk), "causal_attn: the fused kernel is built for T = 256, d_k = 128");
    GNNLM_REQUIRE(ld % 4 == 0 && ((uintptr_t)Q % 16 == 0) && ((uintptr_t)K % 16 == 0) && ((uintptr_t)V % 16 == 0),
                  "causal_attn: operands must be 16-byte aligned with ld % 4 == 0");
    if (n_blocks == 0) return OK;
    constexpr size_t lds_bytes = 2 * 64 * (128 + 8) * sizeof(float);
    GNNLM_LDS_OPT_IN(&causal_attn_256x128_kernel, lds_bytes);
    CausalAttnParams p{Q, K, V, out, ld, ldo, n_blocks, H, max_ctx, accumulate ? 1 : 0};
    const double pairs = (double)n_blocks * H;
    // algorithmic: the causal half of 2 * (T * T * dk) * 2 flops per (block, head)
    ProfScope prof(K_CAUSAL, stream, pairs * 2.0 * T * (T + 1) * dk, pairs * 4.0 * T * dk * 4.0);
    hipLaunchKernelGGL(causal_attn_256x128_kernel, dim3((unsigned)(n_blocks * H)), dim3(512), lds_bytes, stream, p);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

}  // namespace gnnlm
