// Sustained matrix-core rate and shader clock under load on this MI355X: register-only MFMA loops.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak.hip -o gpurun_out/mfma_peak && gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* clk, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
    bf16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(a + i); hb[i] = (__bf16)(b - i); }
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, acc[i], 0, 0, 0);
            }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int MODE>
void run(const char* name, double flops_per_mfma, int wgs_per_cu) {
    const int blocks = 256 * wgs_per_cu, iters = 20000;
    float* out; long long* clk;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double mfmas = (double)blocks * 4 * iters * 16;
    printf("%-28s %d WG/CU: %8.1f TFLOP/s   clock64 delta %lld, wall_clock64 delta %lld (100 MHz) -> %.0f MHz if clock64 counts shader cycles\n",
           name, wgs_per_cu, mfmas * flops_per_mfma / (ms * 1e-3) / 1e12, h[0], h[1], (double)h[0] / h[1] * 100.0);
}
// sustained: ~3 s of back-to-back launches, rate per window
template <int MODE>
void sustained(const char* name, double flops_per_mfma) {
    const int blocks = 512, iters = 20000;
    float* out; long long* clk;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double mfmas = (double)blocks * 4 * iters * 16;
    printf("%s sustained:", name);
    for (int w = 0; w < 6; ++w) {
        hipEventRecord(e0);
        int n = 0;
        for (; n < 60; ++n) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf(" %.1f", n * mfmas * flops_per_mfma / (ms * 1e-3) / 1e12);
    }
    printf(" TFLOP/s\n");
}
int main() {
    sustained<0>("mfma_f32_32x32x2_f32", 32.0 * 32 * 2 * 2);
    sustained<1>("mfma_f32_32x32x16_bf16", 32.0 * 32 * 16 * 2);
    run<0>("mfma_f32_32x32x2_f32", 32.0 * 32 * 2 * 2, 1);
    run<0>("mfma_f32_32x32x2_f32", 32.0 * 32 * 2 * 2, 2);
    run<1>("mfma_f32_32x32x16_bf16", 32.0 * 32 * 16 * 2, 1);
    run<1>("mfma_f32_32x32x16_bf16", 32.0 * 32 * 16 * 2, 2);
    return 0;
}
