// f32 "NT" GEMM for NARROW outputs (N <= 256: the two tail projections of the adaptive softmax, a few hundred live rows x 256
// or 64 columns x K = 1024).  With a dozen 64x64 tiles on 256 CUs such a launch costs the LATENCY of one tile's k-loop (32
// stages of load -> LDS -> barrier -> MFMA, ~30 us), not throughput.
//
// Here a workgroup owns a 32x32 tile and its four waves SPLIT K: wave w walks k in [w K/4, (w + 1) K/4) with the MFMA
// operands loaded straight from global memory into the lane that needs them (lane (l32, half) reads 16 B of row l32 per 8 k:
// no LDS staging, no barrier in the loop, eight groups in flight per wave); the four partial tiles meet in LDS, are summed in
// a fixed order and stored by the whole workgroup with the store epilogue of gemm_epilogue.inc restated per element.
//
// The split changes the summation order of an output element, so WHICH problems take this kernel may not depend on the row
// count: the choice is made from N, K and the batch alone (gemm_skinny_eligible), and a row's result is the same bits whether
// its launch carries one block or thirty-two, a row subset or all rows.
#include "kernels.h"

namespace gnnlm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

__global__ __launch_bounds__(256) void gemm_nt_f32_skinny_kernel(const GemmParams p) {
    constexpr int BT = 32, DEPTH = 8;
    __shared__ float part[4][BT][BT + 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l32 = lane & 31;

    int M = p.M;
    if (p.m_dev) M = min(M, *p.m_dev);
    if (p.m_out && blockIdx.x == 0 && threadIdx.x == 0) *p.m_out = M;
    const int tiles_n = (p.N + BT - 1) / BT;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;      // n fastest: the workgroups of one m-tile share its A rows
    const int m0 = tm * BT, n0 = tn * BT;
    if (m0 >= M) return;

    const int gr = m0 + l32;
    int64_t ar = gr < M ? (p.a_rows ? (int64_t)p.a_rows[gr] : (int64_t)gr) : 0;
    if (ar < 0) ar = 0;                                            // zero row, applied in the epilogue
    const int gn = n0 + l32;
    const int kw = p.K / 4;                                        // this wave's k range (a multiple of 8)
    const float* ap = p.A + ar * p.lda + wave * kw + 4 * half;
    const float* wp = p.W + (int64_t)(gn < p.N ? gn : 0) * p.ldw + wave * kw + 4 * half;
    const int ng = kw / 8;                                         // groups of 8 k: one float4 of A and of W per lane

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 a[DEPTH], b[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        const int gk = d < ng ? d : 0;
        a[d] = *reinterpret_cast<const float4*>(ap + 8 * gk);
        b[d] = *reinterpret_cast<const float4*>(wp + 8 * gk);
    }
    for (int g0 = 0; g0 < ng; g0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (g0 + d < ng) {
                const float4 x = a[d], y = b[d];
                const int gk = g0 + d + DEPTH < ng ? g0 + d + DEPTH : 0;      // the load DEPTH groups ahead (clamped: valid memory)
                a[d] = *reinterpret_cast<const float4*>(ap + 8 * gk);
                b[d] = *reinterpret_cast<const float4*>(wp + 8 * gk);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, y.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, y.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, y.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, y.w, acc, 0, 0, 0);
            }
        }
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * half][l32] = acc[r];
    __syncthreads();

#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = tid + 256 * u, lr = e >> 5, lc = e & 31;
        const int row = m0 + lr, col = n0 + lc;
        if (row >= M || col >= p.N) continue;
        const float s = (part[0][lr][lc] + part[1][lr][lc]) + (part[2][lr][lc] + part[3][lr][lc]);
        const int crow = p.c_rows ? p.c_rows[row] : row;
        float x = (p.a_rows && p.a_rows[row] < 0) ? 0.f : s * p.alpha;
        if (p.bias) x += (p.gate ? p.gate[row] : 1.f) * (p.bias_mode == 1 ? p.bias[col] : p.bias[row]);
        if (p.R) x += p.R[(int64_t)crow * p.ldr + col];
        p.C[(int64_t)crow * p.ldc + col] = x;
    }
}
}  // namespace

// Decided from N, K and the batch only -- never from the row count (see the header).  GNNLM_GEMM_SKINNY=0 switches it off.
bool gemm_skinny_eligible(const GemmParams& p) {
    static const int on = [] { const char* e = getenv("GNNLM_GEMM_SKINNY"); return e ? atoi(e) : 1; }();
    if (!on || p.precision != 0 || p.lse_part) return false;
    if (p.batch1 * p.batch2 != 1 || p.N > 256) return false;
    return p.K % 32 == 0 && p.K >= 512;
}

int gemm_nt_skinny(const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
    const int64_t tiles = cdiv(p.M, 32) * cdiv(p.N, 32);
    GNNLM_REQUIRE(tiles < (1ll << 31), "gemm: grid too large");
    const double work = 2.0 * p.M * (double)p.N * p.K;
    ProfScope prof(K_GEMM, stream, work, 4.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N), p.m_dev, (double)p.M, true);
    if (prof.slot) p.m_out = prof.slot;
    hipLaunchKernelGGL(gemm_nt_f32_skinny_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, p);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

}  // namespace gnnlm
