// f32 "NT" GEMM on the f32 matrix cores with LDS-DMA staging (global_load_lds_dwordx4): the variant of
// gemm_f32.hip's kernel taken for K % BK == 0 problems on 128x128 tiles.  Same contract, same epilogues
// (gemm_epilogue.inc), same numerics (v_mfma_f32_32x32x2_f32 fmaf chains; the k order inside a stage is
// permuted, which the MFMA sums over).
//
// What changes against the register-staged kernel: the operand tiles go global -> LDS directly (no staging
// VGPRs, no ds_write pass, no second barrier per k-tile) and the LDS image is double-buffered with ONE barrier
// per stage.
//
// Measured (tools/gemm_bench.py, TFLOP/s; GNNLM_DMA_EXP ablations built with tools/build_variant.sh):
//   register-staged kernel      head 8192x20002x1024 124.7   4096^3 129.5   163840x1024x1024 127.7
//   this kernel, BK = 16 (4 WG/CU)                   124.0          126.8                    127.1
//   this kernel, BK = 32 (2 WG/CU, the default)      124.5          136.3                    123.4   (step: -1.7 %)
//   BK = 16 without the DMA in the main loop         139.8          147.4      <- LDS reads + MFMA + barrier alone
//   BK = 16, DMA re-reading one L1-resident stage    127.5          130.9      <- the cost is the load issue itself,
//                                                                                 not the L2 / fabric latency
//   register-only MFMA loop (tools/probes/mfma_peak.hip) 155.2 sustained.
// I.e. getting 16 KiB per stage into LDS costs ~10 % of the matrix pipe whichever way it is staged (spreading the
// DMA instructions over the stage's k-step groups instead of issuing them together: no gain either); what helped
// is fewer load instructions per MFMA (256-wide tiles).  Reference point: the vendor library's f32 GEMM
// (tools/vendor_gemm_bench.py) does 131 / 134 / 143 / 149 TFLOP/s on 8192x20002x1024 / 8192x1024x1024 /
// 163840x1024x1024 / 4096^3 -- hand-scheduled assembly hides the loads completely; this HIP kernel gives 5-15 %
// away on the plain shapes and wins where the fused epilogue matters (the head with its log-sum-exp: 2.52 ms here
// against 2.56 ms for the vendor GEMM alone, before any softmax over its 655 MB of logits).
//
// LDS image of a stage: [256 rows (A 128 + W 128)][BK floats], rows unpadded (a DMA instruction writes
// wave-uniform base + lane x 16 B: 64 / (BK/4) consecutive rows).  Bank conflicts are avoided by swizzling on the
// SOURCE side: slot s (16 B) of row r holds the global k-chunk s ^ f(r); a lane's per-lane global address makes
// that free, and the fragment read applies the same involution.  f(r) = (r >> 2) & 3 for 64-B rows (BK = 16):
// the 16 lanes of a ds_read_b128 group (16 consecutive rows, one chunk) then cover all 16 slots of a 256-B
// bank window.
#include "kernels.h"

namespace gnnlm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

namespace {
enum { EPI_STORE = 0, EPI_LSE = 1 };

#ifndef GNNLM_DMA_EXP
#define GNNLM_DMA_EXP 0      // ablations (timing only, wrong results): 1 no DMA after the prologue, 2 + no LDS reads, 3 + no barriers
#endif
// BT = 128: 128x128 tile, 4 waves (2x2) of 64x64.  BT = 256: 256x256 tile, 8 waves (2x4) of 128x64 -- half the
// load instructions and 3/4 of the LDS reads per MFMA, one workgroup (2 waves per SIMD) per CU, 128 KiB of LDS.
template <int EPI, int BK, int BT>
__global__ __launch_bounds__(BT == 256 ? 512 : 256, BT == 256 ? 1 : (BK == 16 ? 4 : 2)) void gemm_nt_f32_dma_kernel(const GemmParams p) {
    constexpr int BM = BT, BN = BT, WAVES_N = BT / 64, NWAVES = 2 * WAVES_N;
    constexpr int WROWS = BM / 2, WCOLS = 64, TM = WROWS / 32, TN = 2;
    constexpr int CH = BK / 4;                     // 16-B chunks per row
    constexpr int RPI = 64 / CH;                   // rows per DMA instruction
    constexpr int NI = 2 * BT / RPI / NWAVES;      // DMA instructions per wave per stage
    constexpr int STAGE = 2 * BT * BK;             // floats
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [2][2 BT rows][BK]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int half = lane >> 5, l32 = lane & 31;

    int M = p.M;
    if (p.m_dev) M = min(M, *p.m_dev);
    if (p.m_out && blockIdx.x == 0 && threadIdx.x == 0) *p.m_out = M;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = ((p.m_dev ? M : p.M) + BM - 1) / BM;
    const unsigned n_tiles = (unsigned)(tiles_m * tiles_n);
    const unsigned n_work = n_tiles * (unsigned)(p.batch1 * p.batch2);
    const int nk = p.K / BK;

    // DMA lane roles: instruction q of this wave stages rows (NI * wave + q) * RPI + lane / CH of the 256-row
    // image, LDS slot lane % CH; the global chunk is slot ^ f(row)
    const int d_slot = lane % CH, d_rsub = lane / CH;
    // fragment reads: lane (l32, half) of k-step group s reads chunk c = 2 s + half of row l32 (+ 32 i)
#define GNNLM_SWZ(row_) (BK == 16 ? (((row_) >> 2) & 3) : (((row_) >> 1) & 7))

    for (unsigned v = blockIdx.x; v < n_work; v += gridDim.x) {
        const unsigned w_ = xcd_remap(v, n_work);
        const unsigned by = w_ / n_tiles, t = w_ - by * n_tiles;
        const int b1 = by / p.batch2, b2 = by % p.batch2;
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
        // log-sum-exp problems: the tile's pick columns are requested now and staged in LDS behind the k-loop -- loaded in the
        // epilogue they cost the whole workgroup a memory round trip per tile
        int pick_reg = -1;
        if (EPI == EPI_LSE && tid < BM && p.lse_pick && m0 + tid < M) pick_reg = p.lse_pick[m0 + tid];
        const float* A = p.A + b1 * p.sA1 + b2 * p.sA2;
        const float* W = p.W + b1 * p.sW1 + b2 * p.sW2;

        const float* src[NI];
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int irow = (NI * wave + q) * RPI + d_rsub;          // row of the 2 BT-row stage image
            const int chunk = d_slot ^ GNNLM_SWZ(irow);
            if (irow < BM) {
                const int gr = m0 + irow;
                int64_t ar = gr < M ? (p.a_rows ? (int64_t)p.a_rows[gr] : (int64_t)gr) : 0;
                if (ar < 0) ar = 0;                                    // zero row, applied in the epilogue
                src[q] = A + ar * p.lda + 4 * chunk;
            } else {
                const int gn = n0 + irow - BM;
                src[q] = W + (int64_t)(gn < p.N ? gn : 0) * p.ldw + 4 * chunk;
            }
        }
#define GNNLM_ISSUE(ks_, buf_)                                                               \
    _Pragma("unroll") for (int q = 0; q < NI; ++q)                                           \
        __builtin_amdgcn_global_load_lds((glb_void_t*)(src[q] + (ks_) * BK),                 \
            (lds_void_t*)(lds + (buf_) * STAGE + (NI * wave + q) * RPI * BK), 16, 0, 0);

        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        GNNLM_ISSUE(0, 0)
        __syncthreads();
        const int arow = wm * WROWS + l32, wrow = BM + wn * WCOLS + l32;
        for (int ks = 0; ks < nk; ++ks) {
            const float* sb = lds + (((GNNLM_DMA_EXP == 2 || GNNLM_DMA_EXP == 3) ? 0 : ks) & 1) * STAGE;
#pragma unroll
            for (int s = 0; s < BK / 8; ++s) {
                float4 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int r = arow + 32 * i;
                    a[i] = *reinterpret_cast<const float4*>(sb + r * BK + 4 * ((2 * s + half) ^ GNNLM_SWZ(r)));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int r = wrow + 32 * j;
                    b[j] = *reinterpret_cast<const float4*>(sb + r * BK + 4 * ((2 * s + half) ^ GNNLM_SWZ(r)));
                }
                if (s == 0 && ks + 1 < nk && (GNNLM_DMA_EXP == 0 || GNNLM_DMA_EXP >= 4)) GNNLM_ISSUE((GNNLM_DMA_EXP == 4 ? 0 : GNNLM_DMA_EXP == 5 ? (ks & 7) : ks + 1), (ks + 1) & 1)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if constexpr (EPI == EPI_LSE) {     // transposed accumulators (gemm_epilogue.inc)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j].x, a[i].x, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j].y, a[i].y, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j].z, a[i].z, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j].w, a[i].w, acc[i][j], 0, 0, 0);
                        } else {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                        }
                    }
            }
            if (GNNLM_DMA_EXP != 3) __syncthreads();            // stage ks+1 landed (the barrier carries the DMA's vmcnt(0)); buffer ks&1 is free
        }
#undef GNNLM_ISSUE

        if (EPI == EPI_LSE) {
            if (tid < BM) reinterpret_cast<int*>(lds)[tid] = pick_reg;          // the k-loop's last barrier freed the buffer
            __syncthreads();
        }
#define GNNLM_LSE_PICK_STAGED
#include "gemm_epilogue.inc"
#undef GNNLM_LSE_PICK_STAGED
        __syncthreads();
    }
#undef GNNLM_SWZ
}

// A-stationary variant for short-K log-sum-exp problems (the 207,744-word tail band: K = 64, ~1000 live rows).
// At K = 64 a 128x128 tile is 2 MFLOP for 64 KiB of operands: the generic kernels are bound by the L2 -> LDS
// path, not by the matrix cores (measured 292 us = 57 % of the f32 MFMA peak).  Here a workgroup keeps ONE m-tile
// for its whole life -- its A operand (128 rows x K) lives in 64 registers per lane -- and walks the n-tiles, so
// only the W tile (32 KiB) moves per 2 MFLOP and only W is read back from LDS.  W tiles arrive by LDS-DMA, double-
// buffered, rows of 256 B swizzled on the source side (slot s of row r holds k-chunk s ^ (r & 15)).
template <int K>
__global__ __launch_bounds__(256, 3) void gemm_lse_astationary_kernel(const GemmParams p) {
    constexpr int EPI = EPI_LSE;
    constexpr int BM = 128, BN = 128, TM = 2, TN = 2, WROWS = 64, WCOLS = 64;
    constexpr int CH = K / 4, RPI = 64 / CH, NI = BN / RPI / 4, TILE = BN * K;
    static_assert(K == 64, "row swizzle below is written for 256-byte rows");
    extern __shared__ __attribute__((aligned(16))) float dyn[];          // [BN][K] W tile, then BM ints
    float* wt = dyn;
    float* lds = dyn + TILE;                                              // the epilogue's pick staging
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l32 = lane & 31;
    const int b1 = 0, b2 = 0;

    int M = p.M;
    if (p.m_dev) M = min(M, *p.m_dev);
    if (p.m_out && blockIdx.x == 0 && threadIdx.x == 0) *p.m_out = M;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (M + BM - 1) / BM;
    if (tiles_m == 0) return;
    const int nslots = (int)gridDim.x / tiles_m;                          // workgroups per m-tile
    if ((int)blockIdx.x >= nslots * tiles_m) return;
    const int tm = blockIdx.x % tiles_m, slot = blockIdx.x / tiles_m;
    const int m0 = tm * BM;

    // the stationary operand: A[row][8 s + 4 half + e] for this lane's two rows
    float4 areg[TM][K / 8];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int gr = m0 + wm * WROWS + 32 * i + l32;
        int64_t ar = gr < M ? (p.a_rows ? (int64_t)p.a_rows[gr] : (int64_t)gr) : 0;
        if (ar < 0) ar = 0;
        const float* arow = p.A + ar * p.lda + 4 * half;
#pragma unroll
        for (int s = 0; s < K / 8; ++s) areg[i][s] = *reinterpret_cast<const float4*>(arow + 8 * s);
    }

    const int d_slot = lane % CH, d_rsub = lane / CH;
#define GNNLM_ISSUE_W(nt_)                                                                   \
    _Pragma("unroll") for (int q = 0; q < NI; ++q) {                                         \
        const int irow = (NI * wave + q) * RPI + d_rsub;                                     \
        const int gn = (nt_) * BN + irow;                                                    \
        const float* src_ = p.W + (int64_t)(gn < p.N ? gn : 0) * p.ldw + 4 * (d_slot ^ (irow & 15)); \
        __builtin_amdgcn_global_load_lds((glb_void_t*)src_,                                  \
            (lds_void_t*)(wt + (NI * wave + q) * RPI * K), 16, 0, 0);                        \
    }

    int nt = slot;
    if (nt >= tiles_n) return;
    GNNLM_ISSUE_W(nt)
    if (tid < BM) reinterpret_cast<int*>(lds)[tid] = (p.lse_pick && m0 + tid < M) ? p.lse_pick[m0 + tid] : -1;   // once: the m-tile is fixed
    __syncthreads();
    // One W buffer: the next tile's DMA is issued when this tile's MFMAs have read it and lands under the
    // log-sum-exp epilogue (~1 us of VALU work), so a second buffer would buy nothing -- and 33 KiB of LDS per
    // workgroup lets three of them share a CU, which is what hides the epilogue behind other waves' MFMAs.
    for (; nt < tiles_n; nt += nslots) {
        const int n0 = nt * BN;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const float* sb = wt;
#pragma unroll
        for (int s = 0; s < K / 8; ++s) {
            float4 b[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wn * WCOLS + 32 * j + l32;
                b[j] = *reinterpret_cast<const float4*>(sb + r * K + 4 * ((2 * s + half) ^ (r & 15)));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {      // operands swapped: transposed accumulators (gemm_epilogue.inc)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j].x, areg[i][s].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j].y, areg[i][s].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j].z, areg[i][s].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j].w, areg[i][s].w, acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();            // every wave has read the tile
        if (nt + nslots < tiles_n) GNNLM_ISSUE_W(nt + nslots)
#define GNNLM_LSE_PICK_STAGED
#include "gemm_epilogue.inc"
#undef GNNLM_LSE_PICK_STAGED
        __syncthreads();            // the next W tile landed (the barrier carries the DMA's vmcnt(0))
    }
#undef GNNLM_ISSUE_W
}
}  // namespace

#ifndef GNNLM_DMA_BK
#define GNNLM_DMA_BK 32
#endif
#ifndef GNNLM_DMA_BIG_TILES
#define GNNLM_DMA_BIG_TILES 2048     // 256x256 tiles from this many of them on (8 per CU); 1 << 30 disables
#endif
#ifndef GNNLM_DMA_MIN_K
#define GNNLM_DMA_MIN_K 128
#endif

bool gemm_dma_eligible(const GemmParams& p) {
#ifdef GNNLM_NO_DMA_GEMM
    return false;
#endif
    // Where it pays, measured per launch inside the step (rocprofv3 trace, us; register-staged -> this kernel):
    // LSE head 8192x20002x1024 2689 -> 2571, tail band 1 147 -> 147; store epilogue: projections 146 -> 145,
    // absorbed queries (K = 128) 191 -> 209, 163840x1024x1024 of the 3-layer path 127.7 -> 123.4 TFLOP/s.  So the
    // LSE problems take this kernel and the store-epilogue problems stay on the register-staged one
    // (GNNLM_DMA_STORE=1 at build time sends them here too, for A/B runs).
#ifndef GNNLM_DMA_STORE
    // ... except the head-sized ones, which get the 256x256 tiles: 655360x1024x1024 (3-layer path) 123 -> 128 TFLOP/s
    // ... and, round 2 (tools/gemm_bench.py step8k, us: register-staged -> this kernel, 128x128 tiles): the step's K = 1024
    // problems 8192x1024x1024 169 -> 157, 8192x3072x1024 (Q, K, V in one) 464 -> 436, Z Wvz (batch 8, N = 128) 150 -> 142;
    // K = 128 (absorbed queries) stays where it is (200 -> 218)
    if (!p.lse_part && (p.m_dev || (p.K < 512 && cdiv(p.M, 256) * cdiv(p.N, 256) * p.batch1 * p.batch2 < GNNLM_DMA_BIG_TILES))) return false;
#endif
    if (p.precision == 0 && p.K == 64 && p.lse_part && p.batch1 * p.batch2 == 1 && p.M <= 128 * 768) return true;   // A-stationary kernel
    return p.precision == 0 && p.K % GNNLM_DMA_BK == 0 && p.K >= GNNLM_DMA_MIN_K;
}

// p is normalised by gemm_nt (the caller checked the problem fills the chip with 128x128 tiles)
template <int EPI, int BK, int BT>
int launch_dma(const GemmParams& p, dim3 grid, hipStream_t stream) {
    constexpr size_t lds_bytes = 2 * (size_t)(2 * BT * BK) * sizeof(float);
    GNNLM_LDS_OPT_IN((&gemm_nt_f32_dma_kernel<EPI, BK, BT>), lds_bytes);      // > 64 KiB of dynamic LDS: once per kernel and device
    hipLaunchKernelGGL((gemm_nt_f32_dma_kernel<EPI, BK, BT>), grid, dim3(BT == 256 ? 512 : 256), lds_bytes, stream, p);
    return OK;
}

int gemm_nt_dma(const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
    constexpr int BK = GNNLM_DMA_BK;
    if (p.K == 64 && p.lse_part) {      // short-K log-sum-exp: A stays in registers, workgroups walk the n-tiles
        constexpr size_t lds_bytes = (128 * 64 + 128) * sizeof(float);
        GNNLM_LDS_OPT_IN(&gemm_lse_astationary_kernel<64>, lds_bytes);
        const double work = 2.0 * p.M * (double)p.N * p.K;
        ProfScope prof(K_GEMM, stream, work, 4.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N),
                       p.m_dev, (double)p.M, true);
        if (prof.slot) p.m_out = prof.slot;
        hipLaunchKernelGGL((gemm_lse_astationary_kernel<64>), dim3(768), dim3(256), lds_bytes, stream, p);
        GNNLM_LAUNCH_CHECK();
        return OK;
    }
    const int64_t nb = (int64_t)p.batch1 * p.batch2;
    const bool big = !p.m_dev && cdiv(p.M, 256) * cdiv(p.N, 256) * nb >= GNNLM_DMA_BIG_TILES;
    const int BT = big ? 256 : 128;
    const int64_t tiles = cdiv(p.M, BT) * cdiv(p.N, BT) * nb;
    GNNLM_REQUIRE(tiles < (1ll << 31), "gemm: grid too large");
    const int64_t pool = 256 * (big ? 1 : (BK == 16 ? 4 : 2));
    dim3 grid((unsigned)std::min<int64_t>(tiles, pool));
    const double work = 2.0 * p.M * (double)p.N * p.K * nb;
    ProfScope prof(K_GEMM, stream, work, 4.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N) * nb,
                   p.m_dev, (double)p.M, true);
    if (prof.slot) p.m_out = prof.slot;
    int rc;
    if (big) rc = p.lse_part ? launch_dma<EPI_LSE, BK, 256>(p, grid, stream) : launch_dma<EPI_STORE, BK, 256>(p, grid, stream);
    else rc = p.lse_part ? launch_dma<EPI_LSE, BK, 128>(p, grid, stream) : launch_dma<EPI_STORE, BK, 128>(p, grid, stream);
    if (rc != OK) return rc;
    GNNLM_LAUNCH_CHECK();
    return OK;
}

}  // namespace gnnlm
