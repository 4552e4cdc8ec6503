// Which dense k does a kept value of v_smfmac_i32_16x16x128_i8's A operand meet?  One kept value = 1 at (row lane m, k block kb, byte s) with index
// position p for its slot; B byte b of lane (n, kbB) carries the id 32 kbB + b (same for every n): D[m][n] = the id it met (0 if it met id 0 or nothing).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
__global__ void one(const v4i* a, const v8i* b, const int* idx, v4i* d) {
    v4i acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_smfmac_i32_16x16x128_i8(a[threadIdx.x], b[threadIdx.x], acc, idx[threadIdx.x], 0, 0);
    d[threadIdx.x] = acc;
}
int main() {
    void *da, *db, *di, *dd;
    hipMalloc(&da, 64 * 16); hipMalloc(&db, 64 * 32); hipMalloc(&di, 64 * 4); hipMalloc(&dd, 64 * 16);
    int8_t Bl[64][32];
    for (int lane = 0; lane < 64; ++lane) for (int b = 0; b < 32; ++b) Bl[lane][b] = (int8_t)(32 * (lane / 16) + b);   // ids 0..127
    hipMemcpy(db, Bl, sizeof(Bl), hipMemcpyHostToDevice);
    for (int kb = 0; kb < 4; ++kb) {
        printf("A lane k-block %d (lane = 16 kb + m, m = 3):\n", kb);
        for (int s = 0; s < 16; ++s) {
            printf("  byte %2d:", s);
            for (int p = 0; p < 4; ++p) {
                int8_t Ac[64][16]; uint32_t idxh[64];
                memset(Ac, 0, sizeof(Ac));
                for (int l = 0; l < 64; ++l) idxh[l] = 0;
                const int lane = 16 * kb + 3;
                Ac[lane][s] = 1;
                // every slot's index = p for slot s; the partner slot of the pair gets another position so that the pair is valid
                for (int l = 0; l < 64; ++l) {
                    uint32_t w = 0;
                    for (int t = 0; t < 16; ++t) w |= (uint32_t)((t & 1) ? 3 : 0) << (2 * t);      // default pair (0, 3)
                    idxh[l] = w;
                }
                uint32_t w = idxh[lane];
                w &= ~(3u << (2 * s)); w |= (uint32_t)p << (2 * s);
                // keep the pair ordered / distinct: partner takes a position different from p
                const int t = s ^ 1; const int q = (p == 3 || (p != 0 && (s & 1))) ? 0 : 3;
                w &= ~(3u << (2 * t)); w |= (uint32_t)((s & 1) ? (p == 0 ? 0 : 0) : (p == 3 ? 3 : 3)) << (2 * t);
                (void)q;
                idxh[lane] = w;
                hipMemcpy(da, Ac, sizeof(Ac), hipMemcpyHostToDevice); hipMemcpy(di, idxh, sizeof(idxh), hipMemcpyHostToDevice);
                hipLaunchKernelGGL(one, dim3(1), dim3(64), 0, 0, (const v4i*)da, (const v8i*)db, (const int*)di, (v4i*)dd);
                int got[64][4];
                hipMemcpy(got, dd, sizeof(got), hipMemcpyDeviceToHost);
                // find the non-zero output: which (lane, r) and value
                int found = 0;
                for (int l = 0; l < 64 && !found; ++l) for (int r = 0; r < 4; ++r) if (got[l][r]) { printf("  p%d -> id %3d at D lane %2d reg %d |", p, got[l][r], l, r); found = 1; break; }
                if (!found) printf("  p%d -> (id 0 or nothing)        |", p);
            }
            printf("\n");
        }
    }
    return 0;
}
