// v_smfmac_i32_16x16x128_i8 on gfx950: operand layout (checked against a host reference; decoded with smfmac_layout.hip) and issue rate
// next to the dense v_mfma_i32_16x16x64_i8.  D[m][n] += sum_k A[m][k] B[k][n], A 16 x 128 with 2:4 structured sparsity along k.
//   B (v8i, 32 B per lane): lane (n = lane % 16, kb = lane / 16) holds B[32 kb + b][n] in byte b
//   A (v4i, 16 B per lane): lane (m = lane % 16, kb = lane / 16) holds 16 KEPT values of row m: bytes 8 h .. 8 h + 7 (h = 0, 1) belong to the
//       16 dense positions k = 32 (2 (kb % 2) + h) + 16 (kb / 2) .. + 15, kept byte s of that half to the group of four 4 (s / 2) .. + 3 in it,
//       at the position idx bits [2 (8 h + s) + 1 : 2 (8 h + s)] name
//   idx (one VGPR per lane): sixteen 2-bit positions
//   D (v4i): lane (n = lane % 16, mg = lane / 16) holds D[4 mg + r][n] in register r
//   hipcc --offload-arch=gfx950 -O3 tools/probes/smfmac_i8.hip -o /tmp/smfmac_i8 && /tmp/smfmac_i8
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));

__global__ void one(const v4i* a, const v8i* b, const int* idx, v4i* d) {
    v4i acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_smfmac_i32_16x16x128_i8(a[threadIdx.x], b[threadIdx.x], acc, idx[threadIdx.x], 0, 0);
    d[threadIdx.x] = acc;
}
template <int SPARSE>
__global__ __launch_bounds__(256) void rate(v4i* out, int iters) {
    v4i a = {(int)threadIdx.x, 2, 3, 4};
    v8i b = {1, 2, 3, 4, 5, 6, 7, (int)blockIdx.x};
    v4i bd = {1, 2, 3, (int)blockIdx.x};
    v4i acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            if (SPARSE) acc[x] = __builtin_amdgcn_smfmac_i32_16x16x128_i8(a, b, acc[x], 0x44444444, 0, 0);
            else acc[x] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bd, acc[x], 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
    // ---- layout check
    int8_t Ad[16][128] = {}, Bd[128][16];
    int8_t Ac[64][16];
    uint32_t idxh[64];
    srand(1);
    for (int k = 0; k < 128; ++k) for (int n = 0; n < 16; ++n) Bd[k][n] = (int8_t)(rand() % 255 - 127);
    for (int lane = 0; lane < 64; ++lane) {
        const int m = lane % 16, kb = lane / 16;
        idxh[lane] = 0;
        for (int g = 0; g < 8; ++g) {                       // kept bytes 2 g, 2 g + 1 of the lane: two values at random positions p0 < p1 of their group of four
            int p0 = rand() % 3, p1 = p0 + 1 + rand() % (3 - p0);
            const int8_t v0 = (int8_t)(rand() % 255 - 127), v1 = (int8_t)(rand() % 255 - 127);
            Ac[lane][2 * g] = v0; Ac[lane][2 * g + 1] = v1;
            idxh[lane] |= (uint32_t)p0 << (4 * g) | (uint32_t)p1 << (4 * g + 2);
            const int base = 32 * (2 * (kb % 2) + g / 4) + 16 * (kb / 2) + 4 * (g % 4);
            Ad[m][base + p0] = v0;
            Ad[m][base + p1] = v1;
        }
    }
    int8_t Bl[64][32];
    for (int lane = 0; lane < 64; ++lane) for (int b = 0; b < 32; ++b) Bl[lane][b] = Bd[32 * (lane / 16) + b][lane % 16];
    int ref[16][16];
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { int s = 0; for (int k = 0; k < 128; ++k) s += (int)Ad[m][k] * Bd[k][n]; ref[m][n] = s; }
    void *da, *db, *di, *dd;
    hipMalloc(&da, 64 * 16); hipMalloc(&db, 64 * 32); hipMalloc(&di, 64 * 4); hipMalloc(&dd, 64 * 16);
    hipMemcpy(da, Ac, 64 * 16, hipMemcpyHostToDevice); hipMemcpy(db, Bl, 64 * 32, hipMemcpyHostToDevice); hipMemcpy(di, idxh, 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(one, dim3(1), dim3(64), 0, 0, (const v4i*)da, (const v8i*)db, (const int*)di, (v4i*)dd);
    int got[64][4];
    hipMemcpy(got, dd, 64 * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 4; ++r) bad += got[lane][r] != ref[4 * (lane / 16) + r][lane % 16];
    printf("layout check: %d of 256 outputs differ from the reference under the layout above%s\n", bad, bad ? "" : "  (layout confirmed)");
    if (bad) for (int lane = 0; lane < 4; ++lane) printf("  lane %d: got %d %d %d %d  want %d %d %d %d\n", lane, got[lane][0], got[lane][1], got[lane][2], got[lane][3],
                                                         ref[0][lane], ref[1][lane], ref[2][lane], ref[3][lane]);
    // ---- rate: 4 independent accumulators per wave, 4 waves per workgroup, one workgroup per SIMD ... 1024 workgroups
    v4i* out; hipMalloc(&out, 1024 * 256 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int sparse = 0; sparse < 2; ++sparse) {
        const int iters = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (sparse) hipLaunchKernelGGL(rate<1>, dim3(1024), dim3(256), 0, 0, out, iters);
            else hipLaunchKernelGGL(rate<0>, dim3(1024), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double insts = 1024.0 * 4 * 4 * iters;
        const double k = sparse ? 128 : 64;
        printf("%s: %.3f ms for %.0f wave instructions = %.1f T dense-equivalent int8 ops/s (16 x 16 x %d x 2 per instruction)\n",
               sparse ? "v_smfmac_i32_16x16x128_i8" : "v_mfma_i32_16x16x64_i8   ", ms, insts, insts * 16 * 16 * k * 2 / (ms * 1e-3) / 1e12, (int)k);
    }
    return 0;
}
