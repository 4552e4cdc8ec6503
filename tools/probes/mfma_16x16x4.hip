// Rate of v_mfma_f32_16x16x4_f32 in the geometry of star_attn_tab_kernel: 512-thread workgroups, ONE per CU (150 KiB of
// dynamic LDS), 4 rotating accumulators per wave, optionally a barrier every 32 MFMAs; plus the SIMD each wave landed on.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_16x16x4.hip -o gpurun_out/mfma16 && gpurun_out/mfma16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int BAR>
__global__ __launch_bounds__(512, 2) void k(float* out, long long* clk, unsigned* hw, int iters) {
    extern __shared__ float lds[];
    f32x4 acc[4];
    f32x16 big[2];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) big[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
    if (threadIdx.x == 0) lds[0] = a;
    __syncthreads();
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                else if (u < 4) big[i & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big[i & 1], 0, 0, 0);   // 16 per iteration: same flops
            }
        if (BAR) __syncthreads();
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += big[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 4) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        hw[blockIdx.x * 8 + threadIdx.x / 64] = id;
        clk[2 + blockIdx.x * 8 + threadIdx.x / 64] = c1 - c0;
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int MODE, int BAR>
void run(const char* name, int blocks) {
    const int iters = 4000;
    float* out; long long* clk; unsigned* hw;
    hipMalloc(&out, blocks * 512 * 4); hipMalloc(&clk, 8 * 64); hipMalloc(&hw, 4 * 64);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, BAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, BAR>), dim3(blocks), dim3(512), 150 * 1024, 0, out, clk, hw, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, BAR>), dim3(blocks), dim3(512), 150 * 1024, 0, out, clk, hw, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[34]; unsigned hh[32];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hh, hw, sizeof(hh), hipMemcpyDeviceToHost);
    const double flops = (double)blocks * 8 * iters * 32 * 2048.0;
    printf("%-34s blocks %5d: %7.1f TFLOP/s  cycles per MFMA-equivalent per SIMD %.1f  (clock64 %lld, wall %lld -> %.0f MHz)\n", name, blocks,
           flops / (ms * 1e-3) / 1e12, (double)h[0] / (iters * 64.0), h[0], h[1], (double)h[0] / h[1] * 100.0);
    printf("    WG0 waves: ");
    for (int w = 0; w < 8; ++w) printf("[simd %u cu %u cyc %lld] ", (hh[w] >> 4) & 3, (hh[w] >> 8) & 15, h[2 + w]);
    printf("\n");
}
int main() {
    run<0, 0>("16x16x4, no barrier", 256);
    run<0, 1>("16x16x4, barrier / 32 MFMAs", 256);
    run<0, 1>("16x16x4, barrier / 32 MFMAs", 2048);
    run<1, 0>("32x32x2, no barrier", 256);
    run<1, 1>("32x32x2, barrier / 16 MFMAs", 2048);
    return 0;
}
