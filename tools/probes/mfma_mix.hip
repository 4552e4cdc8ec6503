// What does an instruction cost next to the f32 MFMA stream of star_attn_tab_kernel?  One workgroup per CU (150 KiB of LDS),
// WPS waves per SIMD; every wave loops over  { 1 MFMA ; NV independent v_add_u32 ; NL independent ds_read_b32 }.
// If VALU / LDS instructions issue in the shadow of the wave's (or its partner's) MFMA the loop costs WPS x 32 (16x16x4)
// or WPS x 8 (4x4x1) cycles per iteration whatever NV, NL are; if they do not, they add.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_mix.hip -o gpurun_out/mfma_mix && gpurun_out/mfma_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int NV, int NL, int SPLIT>
__global__ void k(float* out, long long* clk, int iters) {
    extern __shared__ float lds[];
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
    unsigned v[4] = {threadIdx.x, 1, 2, 3};
    float l[4] = {0, 0, 0, 0};
    const unsigned la = (threadIdx.x & 63) * 4;
    lds[threadIdx.x] = a;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    // SPLIT: waves 0..3 only MFMAs, waves 4.. only the VALU / LDS instructions (one wave of each kind per SIMD)
    const bool do_m = !SPLIT || wave < 4, do_o = !SPLIT || wave >= 4;
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (do_m) {
                if (KIND == 0) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u & 3], 0, 0, 0);
                else acc[u & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[u & 3], 0, 0, 0);
            }
            if (do_o) {
#pragma unroll
                for (int j = 0; j < NV; ++j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[j & 3]) : "v"(la));
#pragma unroll
                for (int j = 0; j < NL; ++j) asm volatile("ds_read_b32 %0, %1" : "=v"(l[j & 3]) : "v"(la));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const long long c1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    for (int i = 0; i < 4; ++i) s += (float)v[i] + l[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) clk[wave] = c1 - c0;
}

template <int KIND, int NV, int NL, int SPLIT>
void run(int wps) {
    const int iters = 2000, blocks = 256, threads = 256 * wps;
    float* out; long long* clk;
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&clk, 16 * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<KIND, NV, NL, SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<KIND, NV, NL, SPLIT>), dim3(blocks), dim3(threads), 150 * 1024, 0, out, clk, iters);
    hipDeviceSynchronize();
    long long h[16];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-8s %s waves/SIMD %d  +%d VALU +%d LDS per MFMA: wave 0 %.1f cycles per step", KIND ? "4x4x1" : "16x16x4", SPLIT ? "split roles," : "every wave, ", wps, NV, NL,
           (double)h[0] / (iters * 16.0));
    if (wps > 1) printf(", wave 4 %.1f", (double)h[4] / (iters * 16.0));
    printf("\n");
    hipFree(out); hipFree(clk);
}
int main() {
    run<0, 0, 0, 0>(1); run<0, 2, 0, 0>(1); run<0, 4, 0, 0>(1); run<0, 0, 2, 0>(1); run<0, 3, 2, 0>(1); run<0, 8, 0, 0>(1);
    run<0, 0, 0, 0>(2); run<0, 3, 2, 0>(2);
    run<0, 3, 2, 1>(2); run<0, 8, 0, 1>(2); run<0, 0, 4, 1>(2);
    run<1, 0, 0, 0>(1); run<1, 1, 0, 0>(1); run<1, 0, 1, 0>(1); run<1, 1, 1, 0>(1); run<1, 2, 1, 0>(1);
    run<1, 0, 0, 0>(2); run<1, 1, 1, 0>(2); run<1, 1, 1, 1>(2); run<1, 2, 0, 1>(2);
    return 0;
}
