// f32-in / f32-accumulate "NT" GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   C[M,N] = alpha * A[M,K] . W[N,K]^T (+ bias) (+ residual)
//
// This is the dense contraction of the HGT message pass (reference: the nn.Linear projections of
// fairseq/models/hgt.py:315-322,401, the relation einsum :347-348, and the adaptive-softmax head /
// tail matmuls of fairseq/modules/adaptive_softmax.py:184-203).  f32 MFMA is bit-for-bit a
// k-ordered fmaf chain, i.e. the reference's fp32 numerics.
//
// Tiling (gfx950): 128x128x32 block tile, 256 threads = 4 waves in a 2x2 grid, each wave owns a
// 64x64 patch = 2x2 MFMA 32x32 accumulators (64 acc VGPRs).  A and W are both K-contiguous, so the
// LDS image is row-major [row][k] with a 36-float row stride: every lane fetches its operands with
// ds_read_b128 (16-lane groups land on 16 distinct 16-B slots: 9*i mod 16 is a bijection).  The k
// index is permuted between the two 32-lane halves (half h of a float4 read covers k = 8s+4h..+3);
// the MFMA sums over k, so A and W only have to agree on the permutation.  Global->LDS staging goes
// through registers (the padded image is not lane-linear, so LDS-DMA cannot write it) with the next
// tile's loads in flight under the current tile's 64 MFMAs; one barrier per k-tile, two LDS buffers.
// blockIdx -> tile mapping is XCD-aware (common.h: xcd_remap) so the 8 column tiles of an A row
// panel run on one XCD's L2.
#include "kernels.h"

namespace gnnlm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int BM = 128, BN = 128, BK = 32, LDS_LD = BK + 4;

__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LDS_LD];
    constexpr int STAGE = (BM + BN) * LDS_LD;     // one k-tile of A followed by one of W

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l32 = lane & 31;

    int M = p.M;
    if (p.m_dev) M = min(M, *p.m_dev);
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m_full = (p.M + BM - 1) / BM;
    const unsigned tile = xcd_remap(blockIdx.x, (unsigned)(tiles_m_full * tiles_n));
    const int m0 = (int)(tile / tiles_n) * BM, n0 = (int)(tile % tiles_n) * BN;
    if (m0 >= M) return;

    const int b1 = blockIdx.y / p.batch2, b2 = blockIdx.y % p.batch2;
    const float* A = p.A + b1 * p.sA1 + b2 * p.sA2;
    const float* W = p.W + b1 * p.sW1 + b2 * p.sW2;

    // per-thread staging rows: 4 rows of A and 4 rows of W, one float4 (k = 4*(tid&7)) each.
    // Out-of-range rows / k are read from a clamped (valid) address and zeroed by a select, so the
    // loads stay unconditional (no branches around them, registers stay registers).
    const int kq = (tid & 7) * 4;
    const int srow = tid >> 3;
    const float *ap0, *ap1, *ap2, *ap3, *wp0, *wp1, *wp2, *wp3;
    bool av0, av1, av2, av3, wv0, wv1, wv2, wv3;
    auto a_ptr = [&](int i, bool& ok) -> const float* {
        const int gr = m0 + srow + 32 * i;
        ok = gr < M;
        int64_t ar = ok ? (p.a_rows ? (int64_t)p.a_rows[gr] : (int64_t)gr) : 0;
        if (ar < 0) { ok = false; ar = 0; }
        return A + ar * p.lda;
    };
    auto w_ptr = [&](int i, bool& ok) -> const float* {
        const int gn = n0 + srow + 32 * i;
        ok = gn < p.N;
        return W + (int64_t)(ok ? gn : 0) * p.ldw;
    };
    ap0 = a_ptr(0, av0); ap1 = a_ptr(1, av1); ap2 = a_ptr(2, av2); ap3 = a_ptr(3, av3);
    wp0 = w_ptr(0, wv0); wp1 = w_ptr(1, wv1); wp2 = w_ptr(2, wv2); wp3 = w_ptr(3, wv3);

    f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }

    float4 ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3;
    const int nk = (p.K + BK - 1) / BK;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

#define GNNLM_LOAD_TILE(kt)                                                              \
    {                                                                                    \
        int k_ = (kt) * BK + kq;                                                         \
        const bool kin_ = k_ < p.K;                                                      \
        k_ = kin_ ? k_ : 0;                                                              \
        ra0 = *reinterpret_cast<const float4*>(ap0 + k_); if (!(av0 && kin_)) ra0 = z4;  \
        ra1 = *reinterpret_cast<const float4*>(ap1 + k_); if (!(av1 && kin_)) ra1 = z4;  \
        ra2 = *reinterpret_cast<const float4*>(ap2 + k_); if (!(av2 && kin_)) ra2 = z4;  \
        ra3 = *reinterpret_cast<const float4*>(ap3 + k_); if (!(av3 && kin_)) ra3 = z4;  \
        rw0 = *reinterpret_cast<const float4*>(wp0 + k_); if (!(wv0 && kin_)) rw0 = z4;  \
        rw1 = *reinterpret_cast<const float4*>(wp1 + k_); if (!(wv1 && kin_)) rw1 = z4;  \
        rw2 = *reinterpret_cast<const float4*>(wp2 + k_); if (!(wv2 && kin_)) rw2 = z4;  \
        rw3 = *reinterpret_cast<const float4*>(wp3 + k_); if (!(wv3 && kin_)) rw3 = z4;  \
    }
#define GNNLM_STORE_TILE(buf)                                                            \
    {                                                                                    \
        float* a_ = &lds[(buf) * STAGE + srow * LDS_LD + kq];                            \
        float* w_ = a_ + BM * LDS_LD;                                                    \
        *reinterpret_cast<float4*>(a_) = ra0;                                            \
        *reinterpret_cast<float4*>(a_ + 32 * LDS_LD) = ra1;                              \
        *reinterpret_cast<float4*>(a_ + 64 * LDS_LD) = ra2;                              \
        *reinterpret_cast<float4*>(a_ + 96 * LDS_LD) = ra3;                              \
        *reinterpret_cast<float4*>(w_) = rw0;                                            \
        *reinterpret_cast<float4*>(w_ + 32 * LDS_LD) = rw1;                              \
        *reinterpret_cast<float4*>(w_ + 64 * LDS_LD) = rw2;                              \
        *reinterpret_cast<float4*>(w_ + 96 * LDS_LD) = rw3;                              \
    }
#define GNNLM_MFMA4(acc, a, b)                                                           \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);                  \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);                  \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);                  \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);

    GNNLM_LOAD_TILE(0);
    GNNLM_STORE_TILE(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) GNNLM_LOAD_TILE(kt + 1);
        const float* a_base = &lds[buf * STAGE + (wm * 64 + l32) * LDS_LD + 4 * half];
        const float* w_base = &lds[buf * STAGE + (BM + wn * 64 + l32) * LDS_LD + 4 * half];
#pragma unroll
        for (int s = 0; s < BK / 8; ++s) {
            const float4 a0 = *reinterpret_cast<const float4*>(a_base + 8 * s);
            const float4 a1 = *reinterpret_cast<const float4*>(a_base + 32 * LDS_LD + 8 * s);
            const float4 b0 = *reinterpret_cast<const float4*>(w_base + 8 * s);
            const float4 b1 = *reinterpret_cast<const float4*>(w_base + 32 * LDS_LD + 8 * s);
            GNNLM_MFMA4(acc00, a0, b0)
            GNNLM_MFMA4(acc01, a0, b1)
            GNNLM_MFMA4(acc10, a1, b0)
            GNNLM_MFMA4(acc11, a1, b1)
        }
        if (kt + 1 < nk) GNNLM_STORE_TILE(buf ^ 1);
        __syncthreads();
    }
#undef GNNLM_LOAD_TILE
#undef GNNLM_STORE_TILE
#undef GNNLM_MFMA4

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* C = p.C + b1 * p.sC1 + b2 * p.sC2;
    const float* bias = p.bias ? p.bias + b1 * p.sB1 + b2 * p.sB2 : nullptr;
    const float* R = p.R ? p.R + b1 * p.sR1 + b2 * p.sR2 : nullptr;
    auto store_patch = [&](const f32x16& acc, int i, int j) {
        const int col = n0 + wn * 64 + j * 32 + l32;
        if (col >= p.N) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row >= M) continue;
            const int64_t crow = p.c_rows ? (int64_t)p.c_rows[row] : (int64_t)row;
            float v = acc[r] * p.alpha;
            if (bias) v += (p.gate ? p.gate[row] : 1.f) * (p.bias_mode == 1 ? bias[col] : bias[row]);
            if (R) v += R[crow * p.ldr + col];
            C[crow * p.ldc + col] = v;
        }
    };
    store_patch(acc00, 0, 0);
    store_patch(acc01, 0, 1);
    store_patch(acc10, 1, 0);
    store_patch(acc11, 1, 1);
}
}  // namespace

int gemm_nt(const GemmParams& desc, hipStream_t stream) {
    GemmParams p = desc;
    if (p.alpha == 0.f) p.alpha = 1.f;
    if (p.batch1 == 0) p.batch1 = 1;
    if (p.batch2 == 0) p.batch2 = 1;
    GNNLM_REQUIRE(p.A && p.W && p.C, "gemm: null operand");
    GNNLM_REQUIRE(p.M >= 0 && p.N > 0 && p.K > 0, "gemm: bad shape");
    GNNLM_REQUIRE(p.K % 4 == 0 && p.lda % 4 == 0 && p.ldw % 4 == 0, "gemm: K, lda, ldw must be multiples of 4");
    GNNLM_REQUIRE(((uintptr_t)p.A % 16 == 0) && ((uintptr_t)p.W % 16 == 0), "gemm: operands must be 16-byte aligned");
    GNNLM_REQUIRE(p.sA1 % 4 == 0 && p.sA2 % 4 == 0 && p.sW1 % 4 == 0 && p.sW2 % 4 == 0, "gemm: batch strides must be multiples of 4");
    GNNLM_REQUIRE(p.batch1 >= 1 && p.batch2 >= 1, "gemm: bad batch");
    GNNLM_REQUIRE(p.precision == 0, "gemm: unknown precision");
    if (p.M == 0) return OK;
    const int64_t tiles = cdiv(p.M, BM) * cdiv(p.N, BN);
    GNNLM_REQUIRE(tiles < (1ll << 31) && (int64_t)p.batch1 * p.batch2 < 65536, "gemm: grid too large");
    dim3 grid((unsigned)tiles, (unsigned)(p.batch1 * p.batch2));
    const double work = 2.0 * p.M * (double)p.N * p.K * p.batch1 * p.batch2;
    ProfScope prof(K_GEMM, stream, work, 4.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N) * p.batch1 * p.batch2,
                   p.m_dev, (double)p.M);
    hipLaunchKernelGGL(gemm_nt_f32_kernel, grid, dim3(256), 0, stream, p);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

}  // namespace gnnlm
