// What does one ADC look-up cost on a CU?  The loop of csrc/ivfpq.hip (v_perm_b32 address, LDS read, packed add; 16 waves per
// CU, 128 KiB of tables, conflict-free addresses) in five variants:
//   0: v_perm + ds_read_b64 + v_pk_add_f32 (the kernel)      1: v_perm + ds_read_b128 + 2 v_pk_add_f32 (four queries per read)
//   2: v_perm + ds_read_b64 (no sums)   3: v_perm + ds_read_b32 + v_add_f32 (one query)   4: ds_read_b64 + v_pk_add_f32 (no v_perm)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_lookup.hip -o gpurun_out/lds_lookup && gpurun_out/lds_lookup
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));



__global__ __launch_bounds__(1024) void k0(float* out, long long* clk, int iters) {
    extern __shared__ float lds[];
    for (int e = threadIdx.x; e < 32768; e += 1024) lds[e] = e * 1e-6f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned w[8], t[16];
    for (int i = 0; i < 8; ++i) w[i] = (threadIdx.x * 2654435761u + i * 40503u) ^ (blockIdx.x * 97u);
    for (int g = 0; g < 16; ++g) {
        if (0 == 1) t[g] = (((lane + 2 * g) & 15) << 4) | (((lane + 2 * g + 1) & 15) << 4) << 8;
        else if (0 == 4) t[g] = ((w[g & 7] >> 8) & 0xff00u) | (((lane + g) & 31) << 3);
        else t[g] = (((lane + 2 * g) & 31) << 3) | (((lane + 2 * g + 1) & 31) << 3) << 8;
    }
    f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
    float a2 = 0.f;
    const unsigned s0 = 0x0c0c0400u, s1 = 0x0c0c0501u;
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        asm volatile(
            "v_perm_b32 v96, %3, %11, %27\n"
            "ds_read_b64 v[96:97], v96\n"
            "v_perm_b32 v100, %3, %11, %28\n"
            "ds_read_b64 v[100:101], v100\n"
            "v_perm_b32 v104, %3, %12, %27\n"
            "ds_read_b64 v[104:105], v104\n"
            "v_perm_b32 v108, %3, %12, %28\n"
            "ds_read_b64 v[108:109], v108\n"
            "v_perm_b32 v112, %4, %13, %27\n"
            "ds_read_b64 v[112:113], v112\n"
            "v_perm_b32 v116, %4, %13, %28\n"
            "ds_read_b64 v[116:117], v116\n"
            "v_perm_b32 v120, %4, %14, %27\n"
            "ds_read_b64 v[120:121], v120\n"
            "v_perm_b32 v124, %4, %14, %28\n"
            "ds_read_b64 v[124:125], v124\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[96:97]\n"
            "v_perm_b32 v96, %5, %15, %27\n"
            "ds_read_b64 v[96:97], v96\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %1, %1, v[100:101]\n"
            "v_perm_b32 v100, %5, %15, %28\n"
            "ds_read_b64 v[100:101], v100\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[104:105]\n"
            "v_perm_b32 v104, %5, %16, %27\n"
            "ds_read_b64 v[104:105], v104\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %1, %1, v[108:109]\n"
            "v_perm_b32 v108, %5, %16, %28\n"
            "ds_read_b64 v[108:109], v108\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[112:113]\n"
            "v_perm_b32 v112, %6, %17, %27\n"
            "ds_read_b64 v[112:113], v112\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %1, %1, v[116:117]\n"
            "v_perm_b32 v116, %6, %17, %28\n"
            "ds_read_b64 v[116:117], v116\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[120:121]\n"
            "v_perm_b32 v120, %6, %18, %27\n"
            "ds_read_b64 v[120:121], v120\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %1, %1, v[124:125]\n"
            "v_perm_b32 v124, %6, %18, %28\n"
            "ds_read_b64 v[124:125], v124\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[96:97]\n"
            "v_perm_b32 v96, %7, %19, %27\n"
            "ds_read_b64 v[96:97], v96\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %1, %1, v[100:101]\n"
            "v_perm_b32 v100, %7, %19, %28\n"
            "ds_read_b64 v[100:101], v100\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[104:105]\n"
            "v_perm_b32 v104, %7, %20, %27\n"
            "ds_read_b64 v[104:105], v104\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %1, %1, v[108:109]\n"
            "v_perm_b32 v108, %7, %20, %28\n"
            "ds_read_b64 v[108:109], v108\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[112:113]\n"
            "v_perm_b32 v112, %8, %21, %27\n"
            "ds_read_b64 v[112:113], v112\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %1, %1, v[116:117]\n"
            "v_perm_b32 v116, %8, %21, %28\n"
            "ds_read_b64 v[116:117], v116\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[120:121]\n"
            "v_perm_b32 v120, %8, %22, %27\n"
            "ds_read_b64 v[120:121], v120\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %1, %1, v[124:125]\n"
            "v_perm_b32 v124, %8, %22, %28\n"
            "ds_read_b64 v[124:125], v124\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[96:97]\n"
            "v_perm_b32 v96, %9, %23, %27\n"
            "ds_read_b64 v[96:97], v96\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %1, %1, v[100:101]\n"
            "v_perm_b32 v100, %9, %23, %28\n"
            "ds_read_b64 v[100:101], v100\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[104:105]\n"
            "v_perm_b32 v104, %9, %24, %27\n"
            "ds_read_b64 v[104:105], v104\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %1, %1, v[108:109]\n"
            "v_perm_b32 v108, %9, %24, %28\n"
            "ds_read_b64 v[108:109], v108\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[112:113]\n"
            "v_perm_b32 v112, %10, %25, %27\n"
            "ds_read_b64 v[112:113], v112\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %1, %1, v[116:117]\n"
            "v_perm_b32 v116, %10, %25, %28\n"
            "ds_read_b64 v[116:117], v116\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[120:121]\n"
            "v_perm_b32 v120, %10, %26, %27\n"
            "ds_read_b64 v[120:121], v120\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %1, %1, v[124:125]\n"
            "v_perm_b32 v124, %10, %26, %28\n"
            "ds_read_b64 v[124:125], v124\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[96:97]\n"
            "s_waitcnt lgkmcnt(6)\n"
            "v_pk_add_f32 %1, %1, v[100:101]\n"
            "s_waitcnt lgkmcnt(5)\n"
            "v_pk_add_f32 %0, %0, v[104:105]\n"
            "s_waitcnt lgkmcnt(4)\n"
            "v_pk_add_f32 %1, %1, v[108:109]\n"
            "s_waitcnt lgkmcnt(3)\n"
            "v_pk_add_f32 %0, %0, v[112:113]\n"
            "s_waitcnt lgkmcnt(2)\n"
            "v_pk_add_f32 %1, %1, v[116:117]\n"
            "s_waitcnt lgkmcnt(1)\n"
            "v_pk_add_f32 %0, %0, v[120:121]\n"
            "s_waitcnt lgkmcnt(0)\n"
            "v_pk_add_f32 %1, %1, v[124:125]\n"
            : "+v"(a0), "+v"(a1), "+v"(a2)
            : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]),
              "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]), "v"(t[5]), "v"(t[6]), "v"(t[7]),
              "v"(t[8]), "v"(t[9]), "v"(t[10]), "v"(t[11]), "v"(t[12]), "v"(t[13]), "v"(t[14]), "v"(t[15]), "s"(s0), "s"(s1)
            : "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",
              "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
    }
    const long long c1 = clock64();
    out[blockIdx.x * 1024 + threadIdx.x] = a0.x + a0.y + a1.x + a1.y + a2;
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}


__global__ __launch_bounds__(1024) void k1(float* out, long long* clk, int iters) {
    extern __shared__ float lds[];
    for (int e = threadIdx.x; e < 32768; e += 1024) lds[e] = e * 1e-6f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned w[8], t[16];
    for (int i = 0; i < 8; ++i) w[i] = (threadIdx.x * 2654435761u + i * 40503u) ^ (blockIdx.x * 97u);
    for (int g = 0; g < 16; ++g) {
        if (1 == 1) t[g] = (((lane + 2 * g) & 15) << 4) | (((lane + 2 * g + 1) & 15) << 4) << 8;
        else if (1 == 4) t[g] = ((w[g & 7] >> 8) & 0xff00u) | (((lane + g) & 31) << 3);
        else t[g] = (((lane + 2 * g) & 31) << 3) | (((lane + 2 * g + 1) & 31) << 3) << 8;
    }
    f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
    float a2 = 0.f;
    const unsigned s0 = 0x0c0c0400u, s1 = 0x0c0c0501u;
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        asm volatile(
            "v_perm_b32 v96, %3, %11, %27\n"
            "ds_read_b128 v[96:99], v96\n"
            "v_perm_b32 v100, %3, %11, %28\n"
            "ds_read_b128 v[100:103], v100\n"
            "v_perm_b32 v104, %3, %12, %27\n"
            "ds_read_b128 v[104:107], v104\n"
            "v_perm_b32 v108, %3, %12, %28\n"
            "ds_read_b128 v[108:111], v108\n"
            "v_perm_b32 v112, %4, %13, %27\n"
            "ds_read_b128 v[112:115], v112\n"
            "v_perm_b32 v116, %4, %13, %28\n"
            "ds_read_b128 v[116:119], v116\n"
            "v_perm_b32 v120, %4, %14, %27\n"
            "ds_read_b128 v[120:123], v120\n"
            "v_perm_b32 v124, %4, %14, %28\n"
            "ds_read_b128 v[124:127], v124\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[96:97]\n"
            "v_pk_add_f32 %1, %1, v[98:99]\n"
            "v_perm_b32 v96, %5, %15, %27\n"
            "ds_read_b128 v[96:99], v96\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[100:101]\n"
            "v_pk_add_f32 %1, %1, v[102:103]\n"
            "v_perm_b32 v100, %5, %15, %28\n"
            "ds_read_b128 v[100:103], v100\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[104:105]\n"
            "v_pk_add_f32 %1, %1, v[106:107]\n"
            "v_perm_b32 v104, %5, %16, %27\n"
            "ds_read_b128 v[104:107], v104\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[108:109]\n"
            "v_pk_add_f32 %1, %1, v[110:111]\n"
            "v_perm_b32 v108, %5, %16, %28\n"
            "ds_read_b128 v[108:111], v108\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[112:113]\n"
            "v_pk_add_f32 %1, %1, v[114:115]\n"
            "v_perm_b32 v112, %6, %17, %27\n"
            "ds_read_b128 v[112:115], v112\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[116:117]\n"
            "v_pk_add_f32 %1, %1, v[118:119]\n"
            "v_perm_b32 v116, %6, %17, %28\n"
            "ds_read_b128 v[116:119], v116\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[120:121]\n"
            "v_pk_add_f32 %1, %1, v[122:123]\n"
            "v_perm_b32 v120, %6, %18, %27\n"
            "ds_read_b128 v[120:123], v120\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[124:125]\n"
            "v_pk_add_f32 %1, %1, v[126:127]\n"
            "v_perm_b32 v124, %6, %18, %28\n"
            "ds_read_b128 v[124:127], v124\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[96:97]\n"
            "v_pk_add_f32 %1, %1, v[98:99]\n"
            "v_perm_b32 v96, %7, %19, %27\n"
            "ds_read_b128 v[96:99], v96\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[100:101]\n"
            "v_pk_add_f32 %1, %1, v[102:103]\n"
            "v_perm_b32 v100, %7, %19, %28\n"
            "ds_read_b128 v[100:103], v100\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[104:105]\n"
            "v_pk_add_f32 %1, %1, v[106:107]\n"
            "v_perm_b32 v104, %7, %20, %27\n"
            "ds_read_b128 v[104:107], v104\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[108:109]\n"
            "v_pk_add_f32 %1, %1, v[110:111]\n"
            "v_perm_b32 v108, %7, %20, %28\n"
            "ds_read_b128 v[108:111], v108\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[112:113]\n"
            "v_pk_add_f32 %1, %1, v[114:115]\n"
            "v_perm_b32 v112, %8, %21, %27\n"
            "ds_read_b128 v[112:115], v112\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[116:117]\n"
            "v_pk_add_f32 %1, %1, v[118:119]\n"
            "v_perm_b32 v116, %8, %21, %28\n"
            "ds_read_b128 v[116:119], v116\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[120:121]\n"
            "v_pk_add_f32 %1, %1, v[122:123]\n"
            "v_perm_b32 v120, %8, %22, %27\n"
            "ds_read_b128 v[120:123], v120\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[124:125]\n"
            "v_pk_add_f32 %1, %1, v[126:127]\n"
            "v_perm_b32 v124, %8, %22, %28\n"
            "ds_read_b128 v[124:127], v124\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[96:97]\n"
            "v_pk_add_f32 %1, %1, v[98:99]\n"
            "v_perm_b32 v96, %9, %23, %27\n"
            "ds_read_b128 v[96:99], v96\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[100:101]\n"
            "v_pk_add_f32 %1, %1, v[102:103]\n"
            "v_perm_b32 v100, %9, %23, %28\n"
            "ds_read_b128 v[100:103], v100\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[104:105]\n"
            "v_pk_add_f32 %1, %1, v[106:107]\n"
            "v_perm_b32 v104, %9, %24, %27\n"
            "ds_read_b128 v[104:107], v104\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[108:109]\n"
            "v_pk_add_f32 %1, %1, v[110:111]\n"
            "v_perm_b32 v108, %9, %24, %28\n"
            "ds_read_b128 v[108:111], v108\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[112:113]\n"
            "v_pk_add_f32 %1, %1, v[114:115]\n"
            "v_perm_b32 v112, %10, %25, %27\n"
            "ds_read_b128 v[112:115], v112\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[116:117]\n"
            "v_pk_add_f32 %1, %1, v[118:119]\n"
            "v_perm_b32 v116, %10, %25, %28\n"
            "ds_read_b128 v[116:119], v116\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[120:121]\n"
            "v_pk_add_f32 %1, %1, v[122:123]\n"
            "v_perm_b32 v120, %10, %26, %27\n"
            "ds_read_b128 v[120:123], v120\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[124:125]\n"
            "v_pk_add_f32 %1, %1, v[126:127]\n"
            "v_perm_b32 v124, %10, %26, %28\n"
            "ds_read_b128 v[124:127], v124\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[96:97]\n"
            "v_pk_add_f32 %1, %1, v[98:99]\n"
            "s_waitcnt lgkmcnt(6)\n"
            "v_pk_add_f32 %0, %0, v[100:101]\n"
            "v_pk_add_f32 %1, %1, v[102:103]\n"
            "s_waitcnt lgkmcnt(5)\n"
            "v_pk_add_f32 %0, %0, v[104:105]\n"
            "v_pk_add_f32 %1, %1, v[106:107]\n"
            "s_waitcnt lgkmcnt(4)\n"
            "v_pk_add_f32 %0, %0, v[108:109]\n"
            "v_pk_add_f32 %1, %1, v[110:111]\n"
            "s_waitcnt lgkmcnt(3)\n"
            "v_pk_add_f32 %0, %0, v[112:113]\n"
            "v_pk_add_f32 %1, %1, v[114:115]\n"
            "s_waitcnt lgkmcnt(2)\n"
            "v_pk_add_f32 %0, %0, v[116:117]\n"
            "v_pk_add_f32 %1, %1, v[118:119]\n"
            "s_waitcnt lgkmcnt(1)\n"
            "v_pk_add_f32 %0, %0, v[120:121]\n"
            "v_pk_add_f32 %1, %1, v[122:123]\n"
            "s_waitcnt lgkmcnt(0)\n"
            "v_pk_add_f32 %0, %0, v[124:125]\n"
            "v_pk_add_f32 %1, %1, v[126:127]\n"
            : "+v"(a0), "+v"(a1), "+v"(a2)
            : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]),
              "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]), "v"(t[5]), "v"(t[6]), "v"(t[7]),
              "v"(t[8]), "v"(t[9]), "v"(t[10]), "v"(t[11]), "v"(t[12]), "v"(t[13]), "v"(t[14]), "v"(t[15]), "s"(s0), "s"(s1)
            : "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",
              "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
    }
    const long long c1 = clock64();
    out[blockIdx.x * 1024 + threadIdx.x] = a0.x + a0.y + a1.x + a1.y + a2;
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}


__global__ __launch_bounds__(1024) void k2(float* out, long long* clk, int iters) {
    extern __shared__ float lds[];
    for (int e = threadIdx.x; e < 32768; e += 1024) lds[e] = e * 1e-6f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned w[8], t[16];
    for (int i = 0; i < 8; ++i) w[i] = (threadIdx.x * 2654435761u + i * 40503u) ^ (blockIdx.x * 97u);
    for (int g = 0; g < 16; ++g) {
        if (2 == 1) t[g] = (((lane + 2 * g) & 15) << 4) | (((lane + 2 * g + 1) & 15) << 4) << 8;
        else if (2 == 4) t[g] = ((w[g & 7] >> 8) & 0xff00u) | (((lane + g) & 31) << 3);
        else t[g] = (((lane + 2 * g) & 31) << 3) | (((lane + 2 * g + 1) & 31) << 3) << 8;
    }
    f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
    float a2 = 0.f;
    const unsigned s0 = 0x0c0c0400u, s1 = 0x0c0c0501u;
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        asm volatile(
            "v_perm_b32 v96, %3, %11, %27\n"
            "ds_read_b64 v[96:97], v96\n"
            "v_perm_b32 v100, %3, %11, %28\n"
            "ds_read_b64 v[100:101], v100\n"
            "v_perm_b32 v104, %3, %12, %27\n"
            "ds_read_b64 v[104:105], v104\n"
            "v_perm_b32 v108, %3, %12, %28\n"
            "ds_read_b64 v[108:109], v108\n"
            "v_perm_b32 v112, %4, %13, %27\n"
            "ds_read_b64 v[112:113], v112\n"
            "v_perm_b32 v116, %4, %13, %28\n"
            "ds_read_b64 v[116:117], v116\n"
            "v_perm_b32 v120, %4, %14, %27\n"
            "ds_read_b64 v[120:121], v120\n"
            "v_perm_b32 v124, %4, %14, %28\n"
            "ds_read_b64 v[124:125], v124\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_perm_b32 v96, %5, %15, %27\n"
            "ds_read_b64 v[96:97], v96\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_perm_b32 v100, %5, %15, %28\n"
            "ds_read_b64 v[100:101], v100\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_perm_b32 v104, %5, %16, %27\n"
            "ds_read_b64 v[104:105], v104\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_perm_b32 v108, %5, %16, %28\n"
            "ds_read_b64 v[108:109], v108\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_perm_b32 v112, %6, %17, %27\n"
            "ds_read_b64 v[112:113], v112\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_perm_b32 v116, %6, %17, %28\n"
            "ds_read_b64 v[116:117], v116\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_perm_b32 v120, %6, %18, %27\n"
            "ds_read_b64 v[120:121], v120\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_perm_b32 v124, %6, %18, %28\n"
            "ds_read_b64 v[124:125], v124\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_perm_b32 v96, %7, %19, %27\n"
            "ds_read_b64 v[96:97], v96\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_perm_b32 v100, %7, %19, %28\n"
            "ds_read_b64 v[100:101], v100\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_perm_b32 v104, %7, %20, %27\n"
            "ds_read_b64 v[104:105], v104\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_perm_b32 v108, %7, %20, %28\n"
            "ds_read_b64 v[108:109], v108\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_perm_b32 v112, %8, %21, %27\n"
            "ds_read_b64 v[112:113], v112\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_perm_b32 v116, %8, %21, %28\n"
            "ds_read_b64 v[116:117], v116\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_perm_b32 v120, %8, %22, %27\n"
            "ds_read_b64 v[120:121], v120\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_perm_b32 v124, %8, %22, %28\n"
            "ds_read_b64 v[124:125], v124\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_perm_b32 v96, %9, %23, %27\n"
            "ds_read_b64 v[96:97], v96\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_perm_b32 v100, %9, %23, %28\n"
            "ds_read_b64 v[100:101], v100\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_perm_b32 v104, %9, %24, %27\n"
            "ds_read_b64 v[104:105], v104\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_perm_b32 v108, %9, %24, %28\n"
            "ds_read_b64 v[108:109], v108\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_perm_b32 v112, %10, %25, %27\n"
            "ds_read_b64 v[112:113], v112\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_perm_b32 v116, %10, %25, %28\n"
            "ds_read_b64 v[116:117], v116\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_perm_b32 v120, %10, %26, %27\n"
            "ds_read_b64 v[120:121], v120\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_perm_b32 v124, %10, %26, %28\n"
            "ds_read_b64 v[124:125], v124\n"
            "s_waitcnt lgkmcnt(7)\n"
            "s_waitcnt lgkmcnt(6)\n"
            "s_waitcnt lgkmcnt(5)\n"
            "s_waitcnt lgkmcnt(4)\n"
            "s_waitcnt lgkmcnt(3)\n"
            "s_waitcnt lgkmcnt(2)\n"
            "s_waitcnt lgkmcnt(1)\n"
            "s_waitcnt lgkmcnt(0)\n"
            : "+v"(a0), "+v"(a1), "+v"(a2)
            : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]),
              "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]), "v"(t[5]), "v"(t[6]), "v"(t[7]),
              "v"(t[8]), "v"(t[9]), "v"(t[10]), "v"(t[11]), "v"(t[12]), "v"(t[13]), "v"(t[14]), "v"(t[15]), "s"(s0), "s"(s1)
            : "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",
              "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
    }
    const long long c1 = clock64();
    out[blockIdx.x * 1024 + threadIdx.x] = a0.x + a0.y + a1.x + a1.y + a2;
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}


__global__ __launch_bounds__(1024) void k3(float* out, long long* clk, int iters) {
    extern __shared__ float lds[];
    for (int e = threadIdx.x; e < 32768; e += 1024) lds[e] = e * 1e-6f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned w[8], t[16];
    for (int i = 0; i < 8; ++i) w[i] = (threadIdx.x * 2654435761u + i * 40503u) ^ (blockIdx.x * 97u);
    for (int g = 0; g < 16; ++g) {
        if (3 == 1) t[g] = (((lane + 2 * g) & 15) << 4) | (((lane + 2 * g + 1) & 15) << 4) << 8;
        else if (3 == 4) t[g] = ((w[g & 7] >> 8) & 0xff00u) | (((lane + g) & 31) << 3);
        else t[g] = (((lane + 2 * g) & 31) << 3) | (((lane + 2 * g + 1) & 31) << 3) << 8;
    }
    f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
    float a2 = 0.f;
    const unsigned s0 = 0x0c0c0400u, s1 = 0x0c0c0501u;
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        asm volatile(
            "v_perm_b32 v96, %3, %11, %27\n"
            "ds_read_b32 v96, v96\n"
            "v_perm_b32 v100, %3, %11, %28\n"
            "ds_read_b32 v100, v100\n"
            "v_perm_b32 v104, %3, %12, %27\n"
            "ds_read_b32 v104, v104\n"
            "v_perm_b32 v108, %3, %12, %28\n"
            "ds_read_b32 v108, v108\n"
            "v_perm_b32 v112, %4, %13, %27\n"
            "ds_read_b32 v112, v112\n"
            "v_perm_b32 v116, %4, %13, %28\n"
            "ds_read_b32 v116, v116\n"
            "v_perm_b32 v120, %4, %14, %27\n"
            "ds_read_b32 v120, v120\n"
            "v_perm_b32 v124, %4, %14, %28\n"
            "ds_read_b32 v124, v124\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_add_f32 %2, %2, v96\n"
            "v_perm_b32 v96, %5, %15, %27\n"
            "ds_read_b32 v96, v96\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_add_f32 %2, %2, v100\n"
            "v_perm_b32 v100, %5, %15, %28\n"
            "ds_read_b32 v100, v100\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_add_f32 %2, %2, v104\n"
            "v_perm_b32 v104, %5, %16, %27\n"
            "ds_read_b32 v104, v104\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_add_f32 %2, %2, v108\n"
            "v_perm_b32 v108, %5, %16, %28\n"
            "ds_read_b32 v108, v108\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_add_f32 %2, %2, v112\n"
            "v_perm_b32 v112, %6, %17, %27\n"
            "ds_read_b32 v112, v112\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_add_f32 %2, %2, v116\n"
            "v_perm_b32 v116, %6, %17, %28\n"
            "ds_read_b32 v116, v116\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_add_f32 %2, %2, v120\n"
            "v_perm_b32 v120, %6, %18, %27\n"
            "ds_read_b32 v120, v120\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_add_f32 %2, %2, v124\n"
            "v_perm_b32 v124, %6, %18, %28\n"
            "ds_read_b32 v124, v124\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_add_f32 %2, %2, v96\n"
            "v_perm_b32 v96, %7, %19, %27\n"
            "ds_read_b32 v96, v96\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_add_f32 %2, %2, v100\n"
            "v_perm_b32 v100, %7, %19, %28\n"
            "ds_read_b32 v100, v100\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_add_f32 %2, %2, v104\n"
            "v_perm_b32 v104, %7, %20, %27\n"
            "ds_read_b32 v104, v104\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_add_f32 %2, %2, v108\n"
            "v_perm_b32 v108, %7, %20, %28\n"
            "ds_read_b32 v108, v108\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_add_f32 %2, %2, v112\n"
            "v_perm_b32 v112, %8, %21, %27\n"
            "ds_read_b32 v112, v112\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_add_f32 %2, %2, v116\n"
            "v_perm_b32 v116, %8, %21, %28\n"
            "ds_read_b32 v116, v116\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_add_f32 %2, %2, v120\n"
            "v_perm_b32 v120, %8, %22, %27\n"
            "ds_read_b32 v120, v120\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_add_f32 %2, %2, v124\n"
            "v_perm_b32 v124, %8, %22, %28\n"
            "ds_read_b32 v124, v124\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_add_f32 %2, %2, v96\n"
            "v_perm_b32 v96, %9, %23, %27\n"
            "ds_read_b32 v96, v96\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_add_f32 %2, %2, v100\n"
            "v_perm_b32 v100, %9, %23, %28\n"
            "ds_read_b32 v100, v100\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_add_f32 %2, %2, v104\n"
            "v_perm_b32 v104, %9, %24, %27\n"
            "ds_read_b32 v104, v104\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_add_f32 %2, %2, v108\n"
            "v_perm_b32 v108, %9, %24, %28\n"
            "ds_read_b32 v108, v108\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_add_f32 %2, %2, v112\n"
            "v_perm_b32 v112, %10, %25, %27\n"
            "ds_read_b32 v112, v112\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_add_f32 %2, %2, v116\n"
            "v_perm_b32 v116, %10, %25, %28\n"
            "ds_read_b32 v116, v116\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_add_f32 %2, %2, v120\n"
            "v_perm_b32 v120, %10, %26, %27\n"
            "ds_read_b32 v120, v120\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_add_f32 %2, %2, v124\n"
            "v_perm_b32 v124, %10, %26, %28\n"
            "ds_read_b32 v124, v124\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_add_f32 %2, %2, v96\n"
            "s_waitcnt lgkmcnt(6)\n"
            "v_add_f32 %2, %2, v100\n"
            "s_waitcnt lgkmcnt(5)\n"
            "v_add_f32 %2, %2, v104\n"
            "s_waitcnt lgkmcnt(4)\n"
            "v_add_f32 %2, %2, v108\n"
            "s_waitcnt lgkmcnt(3)\n"
            "v_add_f32 %2, %2, v112\n"
            "s_waitcnt lgkmcnt(2)\n"
            "v_add_f32 %2, %2, v116\n"
            "s_waitcnt lgkmcnt(1)\n"
            "v_add_f32 %2, %2, v120\n"
            "s_waitcnt lgkmcnt(0)\n"
            "v_add_f32 %2, %2, v124\n"
            : "+v"(a0), "+v"(a1), "+v"(a2)
            : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]),
              "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]), "v"(t[5]), "v"(t[6]), "v"(t[7]),
              "v"(t[8]), "v"(t[9]), "v"(t[10]), "v"(t[11]), "v"(t[12]), "v"(t[13]), "v"(t[14]), "v"(t[15]), "s"(s0), "s"(s1)
            : "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",
              "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
    }
    const long long c1 = clock64();
    out[blockIdx.x * 1024 + threadIdx.x] = a0.x + a0.y + a1.x + a1.y + a2;
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}


__global__ __launch_bounds__(1024) void k4(float* out, long long* clk, int iters) {
    extern __shared__ float lds[];
    for (int e = threadIdx.x; e < 32768; e += 1024) lds[e] = e * 1e-6f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned w[8], t[16];
    for (int i = 0; i < 8; ++i) w[i] = (threadIdx.x * 2654435761u + i * 40503u) ^ (blockIdx.x * 97u);
    for (int g = 0; g < 16; ++g) {
        if (4 == 1) t[g] = (((lane + 2 * g) & 15) << 4) | (((lane + 2 * g + 1) & 15) << 4) << 8;
        else if (4 == 4) t[g] = ((w[g & 7] >> 8) & 0xff00u) | (((lane + g) & 31) << 3);
        else t[g] = (((lane + 2 * g) & 31) << 3) | (((lane + 2 * g + 1) & 31) << 3) << 8;
    }
    f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
    float a2 = 0.f;
    const unsigned s0 = 0x0c0c0400u, s1 = 0x0c0c0501u;
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        asm volatile(
            "ds_read_b64 v[96:97], %11\n"
            "ds_read_b64 v[100:101], %11\n"
            "ds_read_b64 v[104:105], %12\n"
            "ds_read_b64 v[108:109], %12\n"
            "ds_read_b64 v[112:113], %13\n"
            "ds_read_b64 v[116:117], %13\n"
            "ds_read_b64 v[120:121], %14\n"
            "ds_read_b64 v[124:125], %14\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[96:97]\n"
            "ds_read_b64 v[96:97], %15\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %1, %1, v[100:101]\n"
            "ds_read_b64 v[100:101], %15\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[104:105]\n"
            "ds_read_b64 v[104:105], %16\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %1, %1, v[108:109]\n"
            "ds_read_b64 v[108:109], %16\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[112:113]\n"
            "ds_read_b64 v[112:113], %17\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %1, %1, v[116:117]\n"
            "ds_read_b64 v[116:117], %17\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[120:121]\n"
            "ds_read_b64 v[120:121], %18\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %1, %1, v[124:125]\n"
            "ds_read_b64 v[124:125], %18\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[96:97]\n"
            "ds_read_b64 v[96:97], %19\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %1, %1, v[100:101]\n"
            "ds_read_b64 v[100:101], %19\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[104:105]\n"
            "ds_read_b64 v[104:105], %20\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %1, %1, v[108:109]\n"
            "ds_read_b64 v[108:109], %20\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[112:113]\n"
            "ds_read_b64 v[112:113], %21\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %1, %1, v[116:117]\n"
            "ds_read_b64 v[116:117], %21\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[120:121]\n"
            "ds_read_b64 v[120:121], %22\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %1, %1, v[124:125]\n"
            "ds_read_b64 v[124:125], %22\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[96:97]\n"
            "ds_read_b64 v[96:97], %23\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %1, %1, v[100:101]\n"
            "ds_read_b64 v[100:101], %23\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[104:105]\n"
            "ds_read_b64 v[104:105], %24\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %1, %1, v[108:109]\n"
            "ds_read_b64 v[108:109], %24\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[112:113]\n"
            "ds_read_b64 v[112:113], %25\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %1, %1, v[116:117]\n"
            "ds_read_b64 v[116:117], %25\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[120:121]\n"
            "ds_read_b64 v[120:121], %26\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %1, %1, v[124:125]\n"
            "ds_read_b64 v[124:125], %26\n"
            "s_waitcnt lgkmcnt(7)\n"
            "v_pk_add_f32 %0, %0, v[96:97]\n"
            "s_waitcnt lgkmcnt(6)\n"
            "v_pk_add_f32 %1, %1, v[100:101]\n"
            "s_waitcnt lgkmcnt(5)\n"
            "v_pk_add_f32 %0, %0, v[104:105]\n"
            "s_waitcnt lgkmcnt(4)\n"
            "v_pk_add_f32 %1, %1, v[108:109]\n"
            "s_waitcnt lgkmcnt(3)\n"
            "v_pk_add_f32 %0, %0, v[112:113]\n"
            "s_waitcnt lgkmcnt(2)\n"
            "v_pk_add_f32 %1, %1, v[116:117]\n"
            "s_waitcnt lgkmcnt(1)\n"
            "v_pk_add_f32 %0, %0, v[120:121]\n"
            "s_waitcnt lgkmcnt(0)\n"
            "v_pk_add_f32 %1, %1, v[124:125]\n"
            : "+v"(a0), "+v"(a1), "+v"(a2)
            : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]),
              "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]), "v"(t[5]), "v"(t[6]), "v"(t[7]),
              "v"(t[8]), "v"(t[9]), "v"(t[10]), "v"(t[11]), "v"(t[12]), "v"(t[13]), "v"(t[14]), "v"(t[15]), "s"(s0), "s"(s1)
            : "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",
              "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
    }
    const long long c1 = clock64();
    out[blockIdx.x * 1024 + threadIdx.x] = a0.x + a0.y + a1.x + a1.y + a2;
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}

int main() {
    float* o; long long* c;
    hipMalloc(&o, 1024 * 1024 * 4); hipMalloc(&c, 1024 * 8);
    const int iters = 2000, grid = 256;
    void (*ks[5])(float*, long long*, int) = {k0, k1, k2, k3, k4};
    const char* nm[5] = {"perm + b64 + pk_add", "perm + b128 + 2 pk_add", "perm + b64", "perm + b32 + add", "b64 + pk_add"};
    const int qpl[5] = {2, 4, 2, 1, 2};
    for (int v = 0; v < 5; ++v) {
        hipFuncSetAttribute((const void*)ks[v], hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(ks[v], dim3(grid), dim3(1024), 128 * 1024, 0, o, c, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(ks[v], dim3(grid), dim3(1024), 128 * 1024, 0, o, c, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[256]; hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
        const double lookups_per_simd = (double)iters * 32 * 4;      // 4 waves per SIMD
        printf("%-24s %8.3f ms  %6.2f cycles per look-up instruction per SIMD (clock64), %6.2f ns per SIMD look-up, %.2f G query-look-ups/s/CU\n",
               nm[v], ms, h[0] / lookups_per_simd, ms * 1e6 / lookups_per_simd, qpl[v] * 16.0 * iters * 32 * 64 / (ms * 1e6));
    }
    return 0;
}
