// f32-in / f32-accumulate "NT" GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   C[M,N] = alpha * A[M,K] . W[N,K]^T (+ bias) (+ residual)          (epilogue STORE)
//   or, without ever writing C:  per-row log-sum-exp partials + the picked column   (epilogue LSE)
//
// This is the dense contraction of the HGT message pass (reference: the nn.Linear projections of
// fairseq/models/hgt.py:315-322,401, the relation einsum :347-348, and the adaptive-softmax head /
// tail matmuls of fairseq/modules/adaptive_softmax.py:184-203).  f32 MFMA is bit-for-bit a
// k-ordered fmaf chain, i.e. the reference's fp32 numerics.
//
// Tiling (gfx950): BMxBNx32 block tile (128x128 for big problems, 64x64 when 128x128 would leave CUs
// idle), 256 threads = 4 waves in a 2x2 grid, each wave owns (BM/2)x(BN/2) = TMxTN MFMA 32x32
// accumulators.  A and W are both K-contiguous, so the LDS image is row-major [row][k] with a
// 36-float row stride: every lane fetches its operands with ds_read_b128 (16-lane groups land on 16
// distinct 16-B slots: 9*i mod 16 is a bijection).  The k index is permuted between the two 32-lane
// halves (half h of a float4 read covers k = 8s+4h..+3); the MFMA sums over k, so A and W only have
// to agree on the permutation.  Global->LDS staging goes through registers with the next k-tile's loads
// in flight under the current tile's MFMAs; ONE LDS buffer and two barriers per k-tile at 3 waves per SIMD
// measured faster than two buffers at 2 waves per SIMD.  (Problems with K % 32 == 0 on 128x128 tiles
// take gemm_f32_dma.hip instead: same tile, LDS-DMA staging; this kernel keeps the ragged-K, short-K and
// 64x64 cases and the in-kernel split-bf16 modes.)
//
// Work list: the grid is a pool of resident workgroups walking the list of (batch, tile) pairs; list
// position -> pool slot through xcd_remap, so every XCD (block b runs on XCD b % 8, speed only) walks a
// contiguous chunk of the list, ordered so that consecutive tiles share the panel of the LARGER operand
// (m-fastest when W is the big one, n-fastest when A is), which then comes from that XCD's L2.  With a
// device-side row count only the real tiles are on the list.  The first k-tile of a workgroup's next
// tile is loaded under the epilogue of the current one.
//
// LSE epilogue (adaptive softmax): the MFMA operands are swapped, so the accumulators hold the transposed
// tile and a token's 64 logits of a wave's slab sit in registers of two lanes; (max, sum exp) per
// (row, 64-column slab) goes to part[row][slab] -- the logits matrix (164 MB per 2048-token step for the
// WikiText-103 head) is never materialised.  Details in gemm_epilogue.inc.
#include "kernels.h"

namespace gnnlm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int BK = 32, LDS_LD = BK + 4;
#ifndef GNNLM_GEMM_NBUF
#define GNNLM_GEMM_NBUF 1
#endif
constexpr int NBUF = GNNLM_GEMM_NBUF;
enum { EPI_STORE = 0, EPI_LSE = 1 };

// round-to-nearest-even f32 -> bf16 (bit pattern in the low 16 bits)
__device__ __forceinline__ unsigned bf16_rn(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_to_f32(unsigned h) { return __uint_as_float(h << 16); }

// NS = 0: native f32 MFMA (exact fmaf chain).
// NS = 2 / 3: split-bf16 emulation on the bf16 matrix cores (precision 1 / 2 of the C ABI): every f32
//   operand is split while staged into NS bf16 planes (x = x0 + x1 (+ x2)) and the product is assembled from the 3 (NS=2: x0y0, x0y1, x1y0; ~2^-16 relative
//   per product) or 6 (NS=3: adds x0y2, x1y1, x2y0; ~2^-24, i.e. f32-level) cross terms with
//   v_mfma_f32_32x32x16_bf16, f32 accumulate.  3/16 resp. 6/16 of the f32-MFMA cycles per flop.
template <int BM, int BN, int EPI, int NS>
__global__ __launch_bounds__(256, 3) void gemm_nt_f32_kernel(const GemmParams p) {
    constexpr int TM = BM / 64, TN = BN / 64;            // 32x32 accumulators per wave
    constexpr int LA = BM / 32, LW = BN / 32;            // staging float4 per thread
    constexpr int STAGE = (BM + BN) * LDS_LD;
    constexpr int SPLIT_LD = 20;                         // floats (80 B) per bf16 row of 32 k: conflict-free b128 reads
    constexpr int PLANE = (BM + BN) * SPLIT_LD;          // floats per bf16 plane
    constexpr int LDS_FLOATS = NS == 0 ? NBUF * STAGE : NS * PLANE;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l32 = lane & 31;

    int M = p.M;
    if (p.m_dev) M = min(M, *p.m_dev);
    if (p.m_out && blockIdx.x == 0 && threadIdx.x == 0) *p.m_out = M;
    const int tiles_n = (p.N + BN - 1) / BN;
    // Tile walk: the grid is a pool of resident workgroups (gemm_nt sizes it to the chip) walking the list of
    // REAL (batch, tile) pairs -- with a device-side row count a workgroup per worst-case tile would cost
    // ~10 ns of dispatch each, 3x the useful work on the 207,744-word tail band.  Virtual index v -> list
    // position through xcd_remap: every XCD (block b runs on XCD b % 8; the pool size is a multiple of 8)
    // walks its own contiguous chunk of the list, so concurrently running tiles share operand panels in
    // that XCD's L2.  The first k-tile of a workgroup's NEXT tile is loaded before the epilogue of the
    // current one, so short-K problems (K = 64 tail band, K = 128 absorbed queries) do not expose a
    // load latency per tile.
    const int tiles_m = ((p.m_dev ? M : p.M) + BM - 1) / BM;
    const unsigned n_tiles = (unsigned)(tiles_m * tiles_n);
    const unsigned n_work = n_tiles * (unsigned)(p.batch1 * p.batch2);
    unsigned v = blockIdx.x;
    if (v >= n_work) return;

    const int kq = (tid & 7) * 4;
    const int srow = tid >> 3;
    const float* ap[LA];
    const float* wp[LW];
    float4 ra[LA], rw[LW];
    const int nk_full = p.K / BK;              // full k-tiles: loaded without any guard
    const bool k_tail = (p.K % BK) != 0;       // one partial tile, peeled after the main loop
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    int m0x, n0x, b1x, b2x;                    // the tile whose first k-tile is in flight

    // (batch, tile) of list position w_, then the staging row pointers: out-of-range rows read a clamped
    // (valid) address, so the loads stay unconditional.
#define GNNLM_TILE_SETUP(w_)                                                                 \
    {                                                                                        \
        const unsigned by_ = (w_) / n_tiles, t = (w_) - by_ * n_tiles;                       \
        b1x = by_ / p.batch2; b2x = by_ % p.batch2;                                          \
        int tm, tn;                                                                          \
        if (p.tile_order == 1) { tm = t / tiles_n; tn = t % tiles_n; }                       \
        else if (p.tile_order == 2) { tn = t / tiles_m; tm = t % tiles_m; }                  \
        else {   /* bands of GM m-tiles; inside a band n is the slow index */                \
            const int GM = p.tile_order - 2;                                                 \
            const int band = t / (GM * tiles_n);                                             \
            const int m_in = min(GM, tiles_m - band * GM);                                   \
            const int r = t - band * GM * tiles_n;                                           \
            tn = r / m_in;                                                                   \
            tm = band * GM + r % m_in;                                                       \
        }                                                                                    \
        m0x = tm * BM; n0x = tn * BN;                                                        \
        const float* A_ = p.A + b1x * p.sA1 + b2x * p.sA2;                                   \
        const float* W_ = p.W + b1x * p.sW1 + b2x * p.sW2;                                   \
        _Pragma("unroll") for (int i = 0; i < LA; ++i) {                                     \
            const int gr = m0x + srow + 32 * i;                                              \
            int64_t ar = gr < M ? (p.a_rows ? (int64_t)p.a_rows[gr] : (int64_t)gr) : 0;      \
            if (ar < 0) ar = 0;      /* negative gather index = zero row, applied in the epilogue */ \
            ap[i] = A_ + ar * p.lda;                                                         \
        }                                                                                    \
        _Pragma("unroll") for (int i = 0; i < LW; ++i) {                                     \
            const int gn = n0x + srow + 32 * i;                                              \
            wp[i] = W_ + (int64_t)(gn < p.N ? gn : 0) * p.ldw;                               \
        }                                                                                    \
    }

    // No select on the loaded data in the main loop: a select would need the value and so put an
    // s_waitcnt right behind every global load, serialising HBM latency with the 64 MFMAs of the
    // k-tile.  Out-of-range rows read a clamped (valid) address instead; their garbage only reaches
    // output rows / columns that are never stored.
#define GNNLM_LOAD_TILE(kt)                                                                  \
    {                                                                                        \
        const int k_ = (kt) * BK + kq;                                                       \
        _Pragma("unroll") for (int i_ = 0; i_ < LA; ++i_)                                    \
            ra[i_] = *reinterpret_cast<const float4*>(ap[i_] + k_);                          \
        _Pragma("unroll") for (int i_ = 0; i_ < LW; ++i_)                                    \
            rw[i_] = *reinterpret_cast<const float4*>(wp[i_] + k_);                          \
    }
#define GNNLM_LOAD_TILE_GUARDED(kt)                                                          \
    {                                                                                        \
        int k_ = (kt) * BK + kq;                                                             \
        const bool kin_ = k_ < p.K;                                                          \
        k_ = kin_ ? k_ : 0;                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < LA; ++i_) {                                  \
            ra[i_] = *reinterpret_cast<const float4*>(ap[i_] + k_);                          \
            if (!kin_) ra[i_] = z4;                                                          \
        }                                                                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < LW; ++i_) {                                  \
            rw[i_] = *reinterpret_cast<const float4*>(wp[i_] + k_);                          \
            if (!kin_) rw[i_] = z4;                                                          \
        }                                                                                    \
    }
    // Split of 4 consecutive k of one row into NS bf16 planes, 8 bytes per plane.
    //   bf16x6 (NS = 3): every plane is a TRUNCATION (x & 0xFFFF0000: one v_and; the residual x - hi is
    //     exact); three 8-bit pieces cover the 24-bit significand, so x0 + x1 + x2 == x bit for bit and only
    //     the three dropped cross products (<= 2^-24 |a||b| each) separate the result from an f32 product.
    //   bf16x3 (NS = 2): both planes are round-to-nearest (x0 = rn(x), x1 = rn(x - x0)): unbiased, residual
    //     2^-18 |x|; truncation would be cheaper but biases the dropped x1*y1 term (same sign as x*y).
#define GNNLM_PK_TRUNC(lo_, hi_) ((__float_as_uint(hi_) & 0xFFFF0000u) | (__float_as_uint(lo_) >> 16))
#define GNNLM_SPLIT_STORE(v_, row_)                                                          \
    {                                                                                        \
        char* dst_ = reinterpret_cast<char*>(lds) + (row_) * (SPLIT_LD * 4) + kq * 2;        \
        float r0_ = (v_).x, r1_ = (v_).y, r2_ = (v_).z, r3_ = (v_).w;                        \
        if constexpr (NS == 3) {                                                             \
            _Pragma("unroll") for (int pl_ = 0; pl_ < 3; ++pl_) {                            \
                *reinterpret_cast<uint2*>(dst_ + pl_ * (PLANE * 4)) =                        \
                    make_uint2(GNNLM_PK_TRUNC(r0_, r1_), GNNLM_PK_TRUNC(r2_, r3_));          \
                r0_ -= __uint_as_float(__float_as_uint(r0_) & 0xFFFF0000u);                  \
                r1_ -= __uint_as_float(__float_as_uint(r1_) & 0xFFFF0000u);                  \
                r2_ -= __uint_as_float(__float_as_uint(r2_) & 0xFFFF0000u);                  \
                r3_ -= __uint_as_float(__float_as_uint(r3_) & 0xFFFF0000u);                  \
            }                                                                                \
        } else {                                                                             \
            _Pragma("unroll") for (int pl_ = 0; pl_ < 2; ++pl_) {                            \
                const unsigned b0_ = bf16_rn(r0_), b1_ = bf16_rn(r1_), b2_ = bf16_rn(r2_), b3_ = bf16_rn(r3_); \
                *reinterpret_cast<uint2*>(dst_ + pl_ * (PLANE * 4)) = make_uint2(b0_ | (b1_ << 16), b2_ | (b3_ << 16)); \
                r0_ -= __uint_as_float(b0_ << 16);                                           \
                r1_ -= __uint_as_float(b1_ << 16);                                           \
                r2_ -= __uint_as_float(b2_ << 16);                                           \
                r3_ -= __uint_as_float(b3_ << 16);                                           \
            }                                                                                \
        }                                                                                    \
    }
#define GNNLM_STORE_TILE(buf)                                                                \
    if constexpr (NS == 0) {                                                                 \
        float* a_ = &lds[(buf) * STAGE + srow * LDS_LD + kq];                                \
        float* w_ = a_ + BM * LDS_LD;                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < LA; ++i_)                                    \
            *reinterpret_cast<float4*>(a_ + 32 * i_ * LDS_LD) = ra[i_];                      \
        _Pragma("unroll") for (int i_ = 0; i_ < LW; ++i_)                                    \
            *reinterpret_cast<float4*>(w_ + 32 * i_ * LDS_LD) = rw[i_];                      \
    } else {                                                                                 \
        _Pragma("unroll") for (int i_ = 0; i_ < LA; ++i_) GNNLM_SPLIT_STORE(ra[i_], srow + 32 * i_)        \
        _Pragma("unroll") for (int i_ = 0; i_ < LW; ++i_) GNNLM_SPLIT_STORE(rw[i_], BM + srow + 32 * i_)   \
    }
#define GNNLM_SPLIT_MFMA(PA, PB)                                                             \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                           \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                       \
            acc[i][j] = EPI == EPI_LSE ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[PB][j], fa[PA][i], acc[i][j], 0, 0, 0)  \
                                       : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA][i], fb[PB][j], acc[i][j], 0, 0, 0);
#define GNNLM_COMPUTE(buf)                                                                   \
    if constexpr (NS != 0) {                                                                 \
        const char* base_ = reinterpret_cast<const char*>(lds);                              \
        _Pragma("unroll") for (int ks_ = 0; ks_ < BK / 16; ++ks_) {                          \
            bf16x8 fa[NS][TM], fb[NS][TN];                                                   \
            _Pragma("unroll") for (int pl_ = 0; pl_ < NS; ++pl_) {                           \
                _Pragma("unroll") for (int i = 0; i < TM; ++i)                               \
                    fa[pl_][i] = *reinterpret_cast<const bf16x8*>(base_ + (pl_ * (BM + BN) + wm * (BM / 2) + 32 * i + l32) * SPLIT_LD * 4 + 32 * ks_ + 16 * half); \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                               \
                    fb[pl_][j] = *reinterpret_cast<const bf16x8*>(base_ + (pl_ * (BM + BN) + BM + wn * (BN / 2) + 32 * j + l32) * SPLIT_LD * 4 + 32 * ks_ + 16 * half); \
            }                                                                                \
            if constexpr (NS == 3) { GNNLM_SPLIT_MFMA(2, 0) GNNLM_SPLIT_MFMA(1, 1) GNNLM_SPLIT_MFMA(0, 2) }   \
            GNNLM_SPLIT_MFMA(1, 0) GNNLM_SPLIT_MFMA(0, 1) GNNLM_SPLIT_MFMA(0, 0)             \
        }                                                                                    \
    } else {                                                                                 \
        const float* a_base = &lds[(buf) * STAGE + (wm * (BM / 2) + l32) * LDS_LD + 4 * half];            \
        const float* w_base = &lds[(buf) * STAGE + (BM + wn * (BN / 2) + l32) * LDS_LD + 4 * half];       \
        _Pragma("unroll") for (int s_ = 0; s_ < BK / 8; ++s_) {                              \
            float4 a[TM], b[TN];                                                             \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                   \
                a[i] = *reinterpret_cast<const float4*>(a_base + 32 * i * LDS_LD + 8 * s_);  \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                   \
                b[j] = *reinterpret_cast<const float4*>(w_base + 32 * j * LDS_LD + 8 * s_);  \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                   \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) {                             \
                    /* LSE epilogue: operands swapped = transposed accumulator tile (gemm_epilogue.inc) */ \
                    if constexpr (EPI == EPI_LSE) {                                          \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j].x, a[i].x, acc[i][j], 0, 0, 0); \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j].y, a[i].y, acc[i][j], 0, 0, 0); \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j].z, a[i].z, acc[i][j], 0, 0, 0); \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j].w, a[i].w, acc[i][j], 0, 0, 0); \
                    } else {                                                                 \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0); \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0); \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0); \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0); \
                    }                                                                        \
                }                                                                            \
        }                                                                                    \
    }

#define GNNLM_LOAD_FIRST()                                                                   \
    if (nk_full > 0) GNNLM_LOAD_TILE(0) else GNNLM_LOAD_TILE_GUARDED(0)

    GNNLM_TILE_SETUP(xcd_remap(v, n_work))
    GNNLM_LOAD_FIRST()
    for (;;) {
        const int m0 = m0x, n0 = n0x, b1 = b1x, b2 = b2x;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        GNNLM_STORE_TILE(0);
        __syncthreads();
        if (nk_full > 0) {
            for (int kt = 0; kt < nk_full; ++kt) {
                constexpr bool ONE_BUF = NBUF == 1 || NS != 0;
                const int buf = ONE_BUF ? 0 : (kt & 1);
                if (kt + 1 < nk_full) GNNLM_LOAD_TILE(kt + 1);
                GNNLM_COMPUTE(buf);
                if (ONE_BUF) __syncthreads();
                if (kt + 1 < nk_full) { GNNLM_STORE_TILE(ONE_BUF ? 0 : (buf ^ 1)); }
                __syncthreads();
            }
            if (k_tail) {
                GNNLM_LOAD_TILE_GUARDED(nk_full);
                GNNLM_STORE_TILE(0);
                __syncthreads();
                GNNLM_COMPUTE(0);
                __syncthreads();
            }
        } else {            // K < BK: the single guarded tile is the one loaded ahead
            GNNLM_COMPUTE(0);
            __syncthreads();
        }
        v += gridDim.x;
        const bool has_next = v < n_work;
        if (has_next) {
            GNNLM_TILE_SETUP(xcd_remap(v, n_work))
            GNNLM_LOAD_FIRST()
        }
        constexpr int WROWS = BM / 2, WCOLS = BN / 2;
#include "gemm_epilogue.inc"
        __syncthreads();          // the next tile's first k-tile overwrites LDS buffer 0
        if (!has_next) break;
    }
#undef GNNLM_LOAD_FIRST
#undef GNNLM_TILE_SETUP
#undef GNNLM_LOAD_TILE_GUARDED
#undef GNNLM_COMPUTE
#undef GNNLM_SPLIT_MFMA
#undef GNNLM_SPLIT_STORE
#undef GNNLM_PK_TRUNC
#undef GNNLM_LOAD_TILE
#undef GNNLM_STORE_TILE
}

// lse[row] = log sum exp over the row's partial (max, sum) pairs.  WPR waves per row: one for short rows
// (4 rows per workgroup), four for long ones (the 207,744-word tail band has 3,248 pairs per row and only
// ~1000 live rows: a wave per row leaves most of the chip idle).
template <int WPR>
__global__ __launch_bounds__(256) void lse_reduce_kernel(const float2* part, int n_parts, int64_t rows,
                                                         const int32_t* m_dev, float* lse) {
    __shared__ float2 red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = WPR == 1 ? (int64_t)blockIdx.x * 4 + wave : (int64_t)blockIdx.x;
    if (row >= rows || (m_dev && row >= *m_dev)) return;        // uniform per workgroup when WPR == 4
    const float2* pr = part + row * n_parts;
    const int i0 = WPR == 1 ? lane : threadIdx.x, step = 64 * WPR;
    float m = -INFINITY;
    for (int i = i0; i < n_parts; i += step) m = fmaxf(m, pr[i].x);
    m = wave_max(m);
    if (WPR == 4) {
        if (lane == 0) red[wave].x = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[0].x, red[1].x), fmaxf(red[2].x, red[3].x));
    }
    float s = 0.f;
    for (int i = i0; i < n_parts; i += step) {
        const float2 v = pr[i];
        s += v.x == -INFINITY ? 0.f : v.y * expf(v.x - m);
    }
    s = wave_sum(s);
    if (WPR == 4) {
        if (lane == 0) red[wave].y = s;
        __syncthreads();
        s = (red[0].y + red[1].y) + (red[2].y + red[3].y);
    }
    if (threadIdx.x == (WPR == 1 ? 64 * wave : 0)) lse[row] = m + logf(s);
}

template <int BM, int BN, int NS>
void launch_ns(const GemmParams& p, dim3 grid, hipStream_t stream) {
    if constexpr (BN == 128) {          // the LSE epilogue writes one part per 64 columns = one per wave column
        if (p.lse_part) {
            hipLaunchKernelGGL((gemm_nt_f32_kernel<BM, BN, EPI_LSE, NS>), grid, dim3(256), 0, stream, p);
            return;
        }
    }
    hipLaunchKernelGGL((gemm_nt_f32_kernel<BM, BN, EPI_STORE, NS>), grid, dim3(256), 0, stream, p);
}
template <int BM, int BN>
void launch(const GemmParams& p, dim3 grid, hipStream_t stream) {
    if (p.precision == 1) launch_ns<BM, BN, 2>(p, grid, stream);
    else if (p.precision == 2) launch_ns<BM, BN, 3>(p, grid, stream);
    else launch_ns<BM, BN, 0>(p, grid, stream);
}
}  // namespace

thread_local int g_default_gemm_precision = 0;

int gemm_nt(const GemmParams& desc, hipStream_t stream) {
    GemmParams p = desc;
    if (p.precision == 0) p.precision = g_default_gemm_precision;
    if (p.alpha == 0.f) p.alpha = 1.f;
    if (p.batch1 == 0) p.batch1 = 1;
    if (p.batch2 == 0) p.batch2 = 1;
    GNNLM_REQUIRE(p.A && p.W && (p.C || p.lse_part), "gemm: null operand");
    GNNLM_REQUIRE(p.M >= 0 && p.N > 0 && p.K > 0, "gemm: bad shape");
    GNNLM_REQUIRE(p.K % 4 == 0 && p.lda % 4 == 0 && p.ldw % 4 == 0, "gemm: K, lda, ldw must be multiples of 4");
    GNNLM_REQUIRE(((uintptr_t)p.A % 16 == 0) && ((uintptr_t)p.W % 16 == 0), "gemm: operands must be 16-byte aligned");
    GNNLM_REQUIRE(p.sA1 % 4 == 0 && p.sA2 % 4 == 0 && p.sW1 % 4 == 0 && p.sW2 % 4 == 0, "gemm: batch strides must be multiples of 4");
    GNNLM_REQUIRE(p.batch1 >= 1 && p.batch2 >= 1, "gemm: bad batch");
    GNNLM_REQUIRE(p.precision >= 0 && p.precision <= 2, "gemm: precision must be 0 (f32 MFMA), 1 (bf16x3) or 2 (bf16x6)");
    GNNLM_REQUIRE(!p.lse_part || p.batch1 * p.batch2 == 1, "gemm: the LSE epilogue does not support batches");
    GNNLM_REQUIRE(!p.lse_part || p.alpha > 0.f, "gemm: the LSE epilogue needs alpha > 0");
    GNNLM_REQUIRE(p.tile_order >= 0 && p.tile_order <= 66, "gemm: tile_order must be 0 (auto), 1 (n fastest), 2 (m fastest) or 2+GM (bands of GM m-tiles)");
    if (p.M == 0) return OK;
    if (p.tile_order == 0)      // share the larger operand's panel between consecutive tiles
        p.tile_order = (!p.m_dev && (double)p.M > (double)p.N) ? 1 : 2;
    if (gemm_split_eligible(p)) return gemm_nt_split(p, stream);      // pre-split planes + LDS-DMA (gemm_split.hip)
    if (gemm_skinny_eligible(p)) return gemm_nt_skinny(p, stream);    // narrow outputs: 32x32 tiles, k split over the waves (chosen from N, K only)
    const int64_t nb = (int64_t)p.batch1 * p.batch2;
    // 128x128 tiles unless they would leave CUs without a workgroup (256 CUs)
    const int64_t tiles128 = cdiv(p.M, 128) * cdiv(p.N, 128) * nb;
    const bool small = !p.lse_part && tiles128 < 256;
    if (!small && gemm_sched_eligible(p)) return gemm_nt_sched(p, stream);   // hand-placed main loop (gemm_f32_sched.hip)
    if (!small && gemm_dma_eligible(p)) return gemm_nt_dma(p, stream);   // LDS-DMA staged main loop (gemm_f32_dma.hip)
    const int BMN = small ? 64 : 128;
    const int64_t tiles = cdiv(p.M, BMN) * cdiv(p.N, BMN) * nb;
    GNNLM_REQUIRE(tiles < (1ll << 31), "gemm: grid too large");
    // pool of resident workgroups: 256 CUs x the kernel variant's workgroups per CU (a multiple of 8 = XCDs)
    const int64_t pool = 256 * (small ? 4 : (p.precision == 2 ? 2 : 3));
    dim3 grid((unsigned)std::min<int64_t>(tiles, pool));
    const double work = 2.0 * p.M * (double)p.N * p.K * nb;
    ProfScope prof(K_GEMM, stream, work, 4.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N) * nb,
                   p.m_dev, (double)p.M, true);
    if (prof.slot) p.m_out = prof.slot;
    if (small) launch<64, 64>(p, grid, stream);
    else launch<128, 128>(p, grid, stream);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

int lse_reduce(const float* part, int n_parts, int64_t rows, const int32_t* m_dev, float* lse, hipStream_t stream) {
    GNNLM_REQUIRE(part && lse && n_parts > 0, "lse_reduce: bad arguments");
    if (rows == 0) return OK;
    ProfScope prof(K_LSE, stream, 0.0, 8.0 * rows * n_parts, m_dev, (double)rows);
    if (n_parts >= 1024)
        hipLaunchKernelGGL(lse_reduce_kernel<4>, dim3((unsigned)rows), dim3(256), 0, stream,
                           reinterpret_cast<const float2*>(part), n_parts, rows, m_dev, lse);
    else
        hipLaunchKernelGGL(lse_reduce_kernel<1>, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, stream,
                           reinterpret_cast<const float2*>(part), n_parts, rows, m_dev, lse);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

}  // namespace gnnlm
