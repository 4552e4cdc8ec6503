// What can issue next to the int8 MFMA stream of ivfpq_scan8_kernel?  One workgroup per CU (150 KiB of LDS), WPS waves per SIMD;
// every wave loops over tiles of  { NM v_mfma_i32_16x16x64_i8 ; NV independent VALU ; NL independent ds_read_b64 (conflict free) }
// with the three kinds interleaved evenly.  DEP = 1: the MFMAs of a tile read the registers the PREVIOUS tile's ds_reads wrote
// (the look-up -> MFMA dependency of the real kernel).  Prints cycles per tile of wave 0 (x 1 / WPS = per SIMD and tile).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_i8_mix.hip -o gpurun_out/mfma_i8_mix && gpurun_out/mfma_i8_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// FEAT bits: 1 one accumulator chain, 2 look-up addresses come from a v_perm_b32 of a code register (VALU -> LDS address
// dependency; replaces 16 of the NV VALU), 4 compare + ballot + scalar branch on the sums at the end of a tile, 8 a 16-byte
// buffer load per tile, consumed (as the code register) two tiles later
template <int NM, int NV, int NL, int DEP, int FEAT, int U>
__device__ __forceinline__ void steps(i32x4 (&acc)[2], const i32x4& bsel, unsigned (&v)[4], u32x2 (&lw)[16], const u32x2 (&lr)[16], unsigned la, const unsigned (&cw)[4]) {
    const unsigned lp = (la & 0xf8u) | (la >> 8) << 16;             // (lane % 32) * 8 | (lane / 32) << 16
    const unsigned lp16 = (la >> 3 & 15u) * 16u;                    // b128 look-ups: (lane % 16) * 16, 16 slots of 16 B per code row
    constexpr int STEPS = 16;
    if constexpr (U < STEPS) {
        if constexpr (NL * (U + 1) / STEPS > NL * U / STEPS) {
            constexpr int r = NL * U / STEPS;
            if constexpr (FEAT & 2) {
                // {0, half, code byte, lane slot}: one v_perm_b32, as in the scan (32 slots x 8 B per code row: conflict free)
                const unsigned ad = __builtin_amdgcn_perm(cw[(r >> 2) & 3], (FEAT & 64) ? lp16 : lp, ((FEAT & 64) ? 0x0c0c0000u : 0x0c020000u) | (unsigned)(4 + (r & 3)) << 8 | 0u);
                if constexpr (FEAT & 64) {
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    u32x4 t;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(ad));
                    lw[(2 * r) & 15] = u32x2{t[0], t[1]};
                    lw[(2 * r + 1) & 15] = u32x2{t[2], t[3]};
                } else
                asm volatile("ds_read_b64 %0, %1" : "=v"(lw[r & 15]) : "v"(ad));
            } else
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(lw[r & 15]) : "v"(la), "n"(r * 512));
        }
        if constexpr (NM * (U + 1) / STEPS > NM * U / STEPS) {
            constexpr int m = NM * U / STEPS;
            i32x4 a;
            if constexpr (DEP) a = i32x4{(int)lr[(2 * m) & 15][0], (int)lr[(2 * m) & 15][1], (int)lr[(2 * m + 1) & 15][0], (int)lr[(2 * m + 1) & 15][1]};
            else a = bsel;
            acc[(FEAT & 1) ? 0 : (m & 1)] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bsel, acc[(FEAT & 1) ? 0 : (m & 1)], 0, 0, 0);
        }
#pragma unroll
        for (int j = NV * U / STEPS; j < NV * (U + 1) / STEPS; ++j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[j & 3]) : "v"(la));
        __builtin_amdgcn_sched_barrier(0);
        steps<NM, NV, NL, DEP, FEAT, U + 1>(acc, bsel, v, lw, lr, la, cw);
    }
}

template <int NM, int NV, int NL, int DEP, int FEAT>
__global__ void __launch_bounds__(1024) k(int* out, long long* clk, int iters, const unsigned char* codes, int thr) {
    extern __shared__ unsigned char lds[];
    i32x4 acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0;
    i32x4 bsel = {0x01010101, 0, 0x01010101, 0};
    unsigned v[4] = {threadIdx.x, 1, 2, 3};
    u32x2 l0[16], l1[16];
    for (int j = 0; j < 16; ++j) { l0[j] = u32x2{threadIdx.x + j, threadIdx.x * 3}; l1[j] = u32x2{threadIdx.x + j, threadIdx.x * 3 + 1}; }
    const unsigned la = (threadIdx.x & 63) * 8;                       // 64 lanes x 8 B: every bank once per half
    for (int i = threadIdx.x; i < 150 * 256; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = i * 2654435761u;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const long long c0 = clock64();
    unsigned c0w[4] = {threadIdx.x * 2654435761u, threadIdx.x * 40503u, threadIdx.x * 77u, threadIdx.x * 3u}, c1w[4], c2w[4];
    for (int i = 0; i < 4; ++i) { c1w[i] = c0w[i] * 3u; c2w[i] = c0w[i] * 5u; }
    int nsurv = 0, vcnt = 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(codes) + (size_t)blockIdx.x * (1 << 20), 0, 1 << 20, 0x00020000);
    auto tile = [&](u32x2 (&lw)[16], const u32x2 (&lr)[16], const unsigned (&cuse)[4], unsigned (&cload)[4], int t) {
        if constexpr (FEAT & 8) {
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            const v4u x = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(threadIdx.x & 63) * 16, __builtin_amdgcn_readfirstlane(((t * 16 + wave) & 1023) * 1024), 0);
            cload[0] = x.x; cload[1] = x.y; cload[2] = x.z; cload[3] = x.w;
        }
        if constexpr (FEAT & 1) for (int r = 0; r < 4; ++r) acc[0][r] = 0;
        steps<NM, NV, NL, DEP, FEAT, 0>(acc, bsel, v, lw, lr, la, cuse);
        if constexpr (FEAT & 4) {                                      // test right behind the chain, scalar branch on "any survivor"
            const unsigned long long m0 = __builtin_amdgcn_ballot_w64(acc[0][0] >= thr), m1 = __builtin_amdgcn_ballot_w64(acc[0][1] >= thr),
                                     m2 = __builtin_amdgcn_ballot_w64(acc[0][2] >= thr), m3 = __builtin_amdgcn_ballot_w64(acc[0][3] >= thr);
            if ((m0 | m1 | m2 | m3) != 0ull) nsurv += __builtin_popcountll(m0) + __builtin_popcountll(m1) + __builtin_popcountll(m2) + __builtin_popcountll(m3);
        }
        if constexpr (FEAT & 16) {                                     // the same test, no branch: popcounts always
            const unsigned long long m0 = __builtin_amdgcn_ballot_w64(acc[0][0] >= thr), m1 = __builtin_amdgcn_ballot_w64(acc[0][1] >= thr),
                                     m2 = __builtin_amdgcn_ballot_w64(acc[0][2] >= thr), m3 = __builtin_amdgcn_ballot_w64(acc[0][3] >= thr);
            nsurv += __builtin_popcountll(m0) + __builtin_popcountll(m1) + __builtin_popcountll(m2) + __builtin_popcountll(m3);
        }
        if constexpr (FEAT & 32) {                                     // no scalar: per-lane count in a VGPR
#pragma unroll
            for (int r = 0; r < 4; ++r) vcnt += acc[0][r] >= thr ? 1 : 0;
        }
        if (DEP) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    for (int it = 0; it < iters; ++it) {                               // two look-up register sets, three code register sets
        tile(l0, l1, c1w, c0w, 6 * it + 0);
        tile(l1, l0, c2w, c1w, 6 * it + 1);
        tile(l0, l1, c0w, c2w, 6 * it + 2);
        tile(l1, l0, c1w, c0w, 6 * it + 3);
        tile(l0, l1, c2w, c1w, 6 * it + 4);
        tile(l1, l0, c0w, c2w, 6 * it + 5);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long c1 = clock64();
    int sum = 0;
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 4; ++r) sum += acc[i][r];
    for (int i = 0; i < 4; ++i) sum += v[i] + c0w[i] + c1w[i] + c2w[i];
    sum += nsurv + vcnt;
    for (int j = 0; j < 16; ++j) sum += l0[j][0] ^ l0[j][1] ^ l1[j][0] ^ l1[j][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) clk[wave] = c1 - c0;
}

template <int NM, int NV, int NL, int DEP, int FEAT = 0>
void run(int wps) {
    const int iters = 400, blocks = 256, threads = 256 * wps;
    int* out; long long* clk; static unsigned char* codes = nullptr;
    if (!codes) { hipMalloc(&codes, 256u << 20); hipMemset(codes, 0x5a, 256u << 20); }
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&clk, 16 * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<NM, NV, NL, DEP, FEAT>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NM, NV, NL, DEP, FEAT>), dim3(blocks), dim3(threads), 150 * 1024, 0, out, clk, iters, codes, 1 << 30);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<NM, NV, NL, DEP, FEAT>), dim3(blocks), dim3(threads), 150 * 1024, 0, out, clk, iters, codes, 1 << 30);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[16];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const double cyc = (double)h[0] / (iters * 6.0);
    printf("%d waves/SIMD  %2d MFMA + %2d VALU + %2d ds_read_b64 per tile%s%s%s%s%s%s: %.3f us per tile-round (all %d waves), %.0f ticks\n", wps, NM, NV, NL,
           DEP ? " dep" : "", FEAT & 1 ? " 1chain" : "", FEAT & 2 ? " perm-addr" : "", FEAT & 4 ? " cmp+ballot+branch" : FEAT & 16 ? " cmp+ballot" : FEAT & 32 ? " cmp(vgpr)" : "", FEAT & 8 ? " bufload" : "", FEAT & 64 ? " (reads are b128)" : "", ms * 1e3 / (iters * 6.0), wps, cyc);
    hipFree(out); hipFree(clk);
}
int main() {
    // the pipes alone, in pairs, all three
    run<8, 0, 0, 0>(4); run<0, 0, 16, 0>(4); run<0, 21, 0, 0>(4);
    run<8, 0, 16, 0>(4); run<8, 21, 0, 0>(4); run<0, 21, 16, 0>(4); run<8, 21, 16, 0>(4);
    // the scan's structure: MFMAs read the previous tile's look-ups, look-up addresses from v_perm_b32, compare + ballot, code loads
    run<8, 21, 16, 1>(4); run<8, 0, 16, 1, 2>(4); run<8, 5, 16, 1, 2>(4);
    run<8, 5, 16, 1, 7>(4); run<8, 5, 16, 1, 19>(4); run<8, 5, 16, 1, 35>(4);
    run<8, 5, 16, 1, 15>(4); run<8, 5, 16, 1, 10>(4);
    // 16 queries per look-up (ds_read_b128): half the addresses and LDS instructions for the same 128 pairs
    run<8, 0, 8, 1, 66>(4); run<8, 5, 8, 1, 66>(4); run<8, 0, 8, 1, 71>(4);
    // waves per SIMD
    run<8, 5, 16, 1, 7>(3); run<8, 5, 16, 1, 7>(2); run<8, 5, 16, 1, 7>(1);
    return 0;
}
