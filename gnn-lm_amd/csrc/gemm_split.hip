// Split-bf16 emulation of the f32 "NT" GEMM on the CDNA4 bf16 matrix cores, big-tile version
// (precision 1 = bf16x3, 2 = bf16x6 of gnnlm_gemm_t; same contract and epilogues as gemm_f32.hip).
//
// Two kernels per call:
//
// 1. split_planes_kernel: one HBM-bound pass per operand turns the f32 matrix [rows, K] into NS bf16
//    planes (x = x0 + x1 (+ x2)) stored as a TILED IMAGE: 16-byte units (8 consecutive k of one row
//    of one plane), ordered  [64-row block][16-k stage][plane][k half][row in block].  The row gather
//    (a_rows), the device-side row count and the zero padding of rows / k are applied here, so the GEMM
//    main loop has no guards at all.
//
// 2. gemm_planes_kernel: block tile TxT (256x256 with 8 waves, wave tile 128x64; or 128x128 with 4
//    waves, wave tile 64x64), k step = 16 (a stage = 1 step for bf16x6, 2 for bf16x3).  A step of one 64-row block is ONE contiguous chunk of the
//    image (NS x 2 KiB), so staging is pure LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave
//    instruction, destination = wave-uniform base + lane x 16 B, no staging VGPRs and no ds_write) into a
//    double-buffered LDS image [plane][k half][row]: the 32 lanes of an MFMA operand read 32 consecutive
//    16-byte units (conflict-free ds_read_b128 without padding).  One barrier per stage; the DMA of stage
//    s+1 is in flight under the 48 (bf16x6) / 24 (bf16x3) MFMAs per wave of stage s.
//    Per MFMA the kernel reads NS(TM+TN)/(P TM TN) operand fragments from LDS (P = 3 or 6 products) --
//    half (bf16x6) to two thirds (bf16x3) of what a plain bf16 GEMM of the same tiling needs, which is
//    what lets the emulation run closer to the matrix-core peak than a plain bf16 GEMM does.
//
// Accuracy (tests/test_kernels_gpu.py::test_gemm_split_precisions): bf16x6 <= 5e-7 sum|a||b| (the
// native f32 MFMA's level), bf16x3 <= 4e-5.
#include <mutex>
#include <unordered_map>

#include "kernels.h"

namespace gnnlm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

namespace {
enum { EPI_STORE = 0, EPI_LSE = 1 };
constexpr int SK = 16;                                   // k per stage (one 32x32x16 MFMA step)

__device__ __forceinline__ unsigned bf16_rn(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

// NS planes of 8 consecutive k: bf16x6 (NS = 3) by truncation (x0 + x1 + x2 == x exactly), bf16x3
// (NS = 2) by round-to-nearest of the value and of the residual (unbiased, 2^-18 |x| left over).
template <int NS>
__device__ __forceinline__ void split8(float (&v)[8], uint4 (&out)[NS]) {
#pragma unroll
    for (int pl = 0; pl < NS; ++pl) {
        unsigned h[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if constexpr (NS == 3) h[e] = __float_as_uint(v[e]) >> 16;
            else h[e] = bf16_rn(v[e]);
            v[e] -= __uint_as_float(h[e] << 16);
        }
        out[pl] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    }
}

// grid (row blocks, ceil(KS / 8)); 256 threads: 16 lanes cover 128 consecutive k of a row, 16 rows per pass
template <int NS>
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ X, int64_t ld,
                                                           const int32_t* __restrict__ rows_idx, int rows,
                                                           const int32_t* __restrict__ m_dev, int K, int KS,
                                                           uint4* __restrict__ out) {
    const int kc = threadIdx.x & 15, rsub = threadIdx.x >> 4;
    const int ks = blockIdx.y * 8 + (kc >> 1), h = kc & 1;
    if (ks >= KS) return;
    const int k = ks * SK + 8 * h;
    int M = rows;
    if (m_dev) M = min(M, *m_dev);
    const int64_t rb = blockIdx.x;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int r = rsub + 16 * pass;
        const int64_t g = rb * 64 + r;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int64_t src = g < M ? (rows_idx ? (int64_t)rows_idx[g] : g) : -1;
        if (src >= 0) {
            const float* px = X + src * ld + k;
            if (k < K) { const float4 a = *reinterpret_cast<const float4*>(px); v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; }
            if (k + 4 < K) { const float4 b = *reinterpret_cast<const float4*>(px + 4); v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; }
        }
        uint4 pk[NS];
        split8<NS>(v, pk);
#pragma unroll
        for (int pl = 0; pl < NS; ++pl)
            out[((((rb * KS + ks) * NS + pl) * 2 + h) << 6) + r] = pk[pl];
    }
}

template <int T, int NS, int EPI>
__global__ __launch_bounds__(T == 256 ? 512 : 256, T == 256 ? 1 : 2)
void gemm_planes_kernel(const GemmParams p, const uint4* __restrict__ Ap, const uint4* __restrict__ Wp, const int KS) {
    constexpr int BM = T, BN = T;
    constexpr int WAVES_N = T / 64;                      // T = 256: 2 x 4 waves of 128 x 64; T = 128: 2 x 2 of 64 x 64
    constexpr int WROWS = BM / 2, WCOLS = 64;
    constexpr int TM = WROWS / 32, TN = WCOLS / 32;
    constexpr int RB = T / 64;                           // 64-row blocks per operand per tile
    constexpr int PH = 2 * NS;                           // (plane, k half) slabs of a 16-k step
    constexpr int OPU = PH * T;                          // 16-B units per operand per 16-k step
    constexpr int KSTEPS = NS == 2 ? 2 : 1;              // 16-k steps per stage (bf16x3: 2, so that a stage is 48 MFMAs too)
    constexpr int STAGE_U = KSTEPS * 2 * OPU;            // units per stage
    static_assert(2 * RB == 2 * WAVES_N, "one (operand, row block) per wave");
    __shared__ uint4 lds[2][STAGE_U];                    // [buffer][k step][operand][plane][k half][row]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: LDS-DMA bases stay in SGPRs
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int half = lane >> 5, l32 = lane & 31;
    const int b1 = 0, b2 = 0;
    const int late = wave >= WAVES_N ? 1 : 0;

    int M = p.M;
    if (p.m_dev) M = min(M, *p.m_dev);
    if (p.m_out && blockIdx.x == 0 && threadIdx.x == 0) *p.m_out = M;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = ((p.m_dev ? M : p.M) + BM - 1) / BM;
    const unsigned n_tiles = (unsigned)(tiles_m * tiles_n);
    const unsigned t_step = p.m_dev ? gridDim.x : n_tiles;
    for (unsigned t = p.m_dev ? blockIdx.x : xcd_remap(blockIdx.x, n_tiles); t < n_tiles; t += t_step) {
    int tm, tn;
    if (p.tile_order == 1) { tm = t / tiles_n; tn = t % tiles_n; }
    else if (p.tile_order == 2) { tn = t / tiles_m; tm = t % tiles_m; }
    else {
        const int GM = p.tile_order - 2;
        const int band = t / (GM * tiles_n);
        const int m_in = min(GM, tiles_m - band * GM);
        const int r = t - band * GM * tiles_n;
        tn = r / m_in;
        tm = band * GM + r % m_in;
    }
    const int m0 = tm * BM, n0 = tn * BN;

    // this wave's share of every stage: the PH slabs of one 64-row block of one operand (PH KiB, contiguous)
    const int s_op = wave / RB, s_rb = wave % RB;
    const uint4* gsrc = (s_op == 0 ? Ap + (int64_t)(tm * RB + s_rb) * KS * (PH * 64)
                                   : Wp + (int64_t)(tn * RB + s_rb) * KS * (PH * 64)) + lane;
    const int s_dst = s_op * OPU + s_rb * 64;

    // two of this wave's PH slabs of 16-k step ks_ -> LDS buffer buf_, step slot st_ (slabs q_, q_+1)
#ifndef GNNLM_EXP
#define GNNLM_EXP 0
#endif
#define GNNLM_ISSUE2(ks_, buf_, st_, q_)                                                     \
    if ((ks_) < KS && !(GNNLM_EXP == 2 && (ks_) > 1)) {                                                                        \
        const uint4* g_ = gsrc + (int64_t)(ks_) * (PH * 64) + (q_) * 64;                     \
        __builtin_amdgcn_global_load_lds((glb_void_t*)(g_), (lds_void_t*)(&lds[buf_][(st_) * 2 * OPU + s_dst + (q_) * T]), 16, 0, 0);            \
        __builtin_amdgcn_global_load_lds((glb_void_t*)(g_ + 64), (lds_void_t*)(&lds[buf_][(st_) * 2 * OPU + s_dst + ((q_) + 1) * T]), 16, 0, 0); \
    }
    // DMA pair q of the next stage goes out after product group q (waves 0..n/2-1) or q+1 (the other half):
    // the two waves of a SIMD never pay the ~100-cycle LDS-DMA issue cost at the same time, so one of them
    // always feeds the matrix pipe.
#define GNNLM_ISSUE_SLOT(g_)                                                                 \
    {                                                                                        \
        const int q_ = (g_) - late;                                                          \
        if (q_ >= 0 && q_ < KSTEPS * NS)                                                     \
            GNNLM_ISSUE2(ks + KSTEPS + q_ / NS, buf ^ 1, q_ / NS, 2 * (q_ % NS))             \
    }
#define GNNLM_READ_A(pl_)                                                                    \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                           \
        fa[pl_][i] = *reinterpret_cast<const bf16x8*>(&lds[buf][st * 2 * OPU + a_off + (pl_) * 2 * T + 32 * i]);
#define GNNLM_READ_W(pl_)                                                                    \
    _Pragma("unroll") for (int j = 0; j < TN; ++j)                                           \
        fb[pl_][j] = *reinterpret_cast<const bf16x8*>(&lds[buf][st * 2 * OPU + w_off + (pl_) * 2 * T + 32 * j]);
#define GNNLM_SPLIT_MFMA(PA, PB)                                                             \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                           \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                       \
            acc[i][j] = EPI == EPI_LSE ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[PB][j], fa[PA][i], acc[i][j], 0, 0, 0)  \
                                       : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA][i], fb[PB][j], acc[i][j], 0, 0, 0);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
    for (int st = 0; st < KSTEPS; ++st) {
        GNNLM_ISSUE2(st, 0, st, 0) GNNLM_ISSUE2(st, 0, st, 2)
        if constexpr (NS == 3) GNNLM_ISSUE2(st, 0, st, 4)
    }
    __syncthreads();                                     // carries the vmcnt(0) of the DMA
    const int a_off = half * T + wm * WROWS + l32;
    const int w_off = OPU + half * T + wn * WCOLS + l32;
    for (int ks = 0; ks < KS; ks += KSTEPS) {            // KS is a multiple of KSTEPS (zero-padded image)
        const int buf = (ks / KSTEPS) & 1;
        // The first product group needs only plane 0 of both operands (TM + TN reads); the other planes
        // land under its MFMAs, and the DMA of the next stage is issued between the groups (an LDS-DMA
        // instruction costs the wave ~100 issue cycles, hidden behind the matrix pipe this way).  The
        // order of the product groups is irrelevant for accuracy: the accumulator already holds the
        // large terms of the earlier k.
#pragma unroll
        for (int st = 0; st < KSTEPS; ++st) {
            bf16x8 fa[NS][TM], fb[NS][TN];
            GNNLM_READ_A(0) GNNLM_READ_W(0) GNNLM_READ_A(1) GNNLM_READ_W(1)
            if constexpr (NS == 3) { GNNLM_READ_A(2) GNNLM_READ_W(2) }
            GNNLM_SPLIT_MFMA(0, 0)
            GNNLM_ISSUE_SLOT(st * (NS == 3 ? 6 : 3) + 0)
            GNNLM_SPLIT_MFMA(1, 0)
            GNNLM_ISSUE_SLOT(st * (NS == 3 ? 6 : 3) + 1)
            GNNLM_SPLIT_MFMA(0, 1)
            GNNLM_ISSUE_SLOT(st * (NS == 3 ? 6 : 3) + 2)
            if constexpr (NS == 3) {
                GNNLM_SPLIT_MFMA(2, 0)
                GNNLM_ISSUE_SLOT(3)
                GNNLM_SPLIT_MFMA(0, 2)
                GNNLM_ISSUE_SLOT(4)
                GNNLM_SPLIT_MFMA(1, 1)
            }
        }
        __syncthreads();                                 // next stage landed; buffer `buf` is free for the one after
    }
#undef GNNLM_SPLIT_MFMA
#undef GNNLM_READ_A
#undef GNNLM_READ_W
#undef GNNLM_ISSUE_SLOT
#undef GNNLM_ISSUE2

#include "gemm_epilogue.inc"
    __syncthreads();
    }   // tile walk
}

// grow-only scratch for the plane images, one per stream (launches on one stream are ordered)
struct Scratch { void* ptr = nullptr; size_t bytes = 0; };
std::mutex g_scratch_mu;
std::unordered_map<hipStream_t, Scratch> g_scratch;

int scratch_get(hipStream_t stream, size_t bytes, void** out) {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    Scratch& s = g_scratch[stream];
    if (s.bytes < bytes) {
        if (s.ptr) {
            GNNLM_HIP(hipStreamSynchronize(stream));
            GNNLM_HIP(hipFree(s.ptr));
            s.ptr = nullptr; s.bytes = 0;
        }
        const size_t want = bytes + bytes / 8;
        GNNLM_HIP(hipMalloc(&s.ptr, want));
        s.bytes = want;
    }
    *out = s.ptr;
    return OK;
}

template <int NS>
void launch_split(const float* X, int64_t ld, const int32_t* rows_idx, int rows, const int32_t* m_dev, int K, int KS,
                  int n_rb, uint4* out, hipStream_t stream) {
    hipLaunchKernelGGL((split_planes_kernel<NS>), dim3((unsigned)n_rb, (unsigned)cdiv(KS, 8)), dim3(256), 0, stream,
                       X, ld, rows_idx, rows, m_dev, K, KS, out);
}

template <int T, int NS>
void launch_planes(const GemmParams& p, const uint4* Ap, const uint4* Wp, int KS, dim3 grid, hipStream_t stream) {
    const dim3 block(T == 256 ? 512 : 256);
    if (p.lse_part)
        hipLaunchKernelGGL((gemm_planes_kernel<T, NS, EPI_LSE>), grid, block, 0, stream, p, Ap, Wp, KS);
    else
        hipLaunchKernelGGL((gemm_planes_kernel<T, NS, EPI_STORE>), grid, block, 0, stream, p, Ap, Wp, KS);
}
}  // namespace

bool gemm_split_eligible(const GemmParams& p) {
    return p.precision != 0 && p.batch1 * p.batch2 == 1 && p.K >= 256 &&
           cdiv(p.M, 128) * cdiv(p.N, 128) >= 256;
}

// p is normalised by gemm_nt (alpha, batch, tile_order resolved)
int gemm_nt_split(const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
    const int NS = p.precision == 1 ? 2 : 3;
    // 256x256 tiles when they still give every CU >= 2 workgroups over the launch.  A device-side M is a small fraction of its bound
    // where the bound is a token count (the softmax tails: 128-tiles), and of the order of the bound where it counts the slot rows
    // of a batch's context groups (ABI 9: the ntgt projections of a multi-layer step, M >= 2^17 -- merged groups are a good
    // third of all groups or more): those keep the big tiles (round 5: 22.2 k -> 25 k tokens/s on the 3-layer recipe under bf16x3)
    const bool big = (!p.m_dev || p.M >= (1 << 17)) && cdiv(p.M, 256) * cdiv(p.N, 256) >= 512;
    const int T = big ? 256 : 128;
    const int KS = (int)cdiv(p.K, SK * 2) * 2;           // 16-k steps, zero-padded to a whole number of stages
    const int64_t rbA = cdiv(p.M, T) * (T / 64), rbW = cdiv(p.N, T) * (T / 64);
    const size_t unit_bytes = 16, per_rb = (size_t)KS * NS * 2 * 64 * unit_bytes;
    void* ws = nullptr;
    const int rc = scratch_get(stream, (size_t)(rbA + rbW) * per_rb, &ws);
    if (rc != OK) return rc;
    uint4* Ap = reinterpret_cast<uint4*>(ws);
    uint4* Wp = Ap + (size_t)rbA * per_rb / unit_bytes;
    {
        ProfScope prof(K_SPLIT, stream, 0.0, (4.0 + 2.0 * NS) * ((double)p.M + (double)p.N) * p.K);
        if (NS == 2) {
            launch_split<2>(p.A, p.lda, p.a_rows, p.M, p.m_dev, p.K, KS, (int)rbA, Ap, stream);
            launch_split<2>(p.W, p.ldw, nullptr, p.N, nullptr, p.K, KS, (int)rbW, Wp, stream);
        } else {
            launch_split<3>(p.A, p.lda, p.a_rows, p.M, p.m_dev, p.K, KS, (int)rbA, Ap, stream);
            launch_split<3>(p.W, p.ldw, nullptr, p.N, nullptr, p.K, KS, (int)rbW, Wp, stream);
        }
        GNNLM_LAUNCH_CHECK();
    }
    const int64_t tiles = cdiv(p.M, T) * cdiv(p.N, T);
    GNNLM_REQUIRE(tiles < (1ll << 31), "gemm: grid too large");
    dim3 grid((unsigned)(p.m_dev ? std::min<int64_t>(tiles, 512) : tiles));
    const double work = 2.0 * p.M * (double)p.N * p.K;
    ProfScope prof(K_GEMM, stream, work, 4.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N), p.m_dev, (double)p.M, true);
    if (prof.slot) p.m_out = prof.slot;
    if (big) { if (NS == 2) launch_planes<256, 2>(p, Ap, Wp, KS, grid, stream); else launch_planes<256, 3>(p, Ap, Wp, KS, grid, stream); }
    else     { if (NS == 2) launch_planes<128, 2>(p, Ap, Wp, KS, grid, stream); else launch_planes<128, 3>(p, Ap, Wp, KS, grid, stream); }
    GNNLM_LAUNCH_CHECK();
    return OK;
}

}  // namespace gnnlm
