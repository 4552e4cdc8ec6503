// What does rocprofv3's FETCH_SIZE report for accesses narrower than a cache line?  Three kernels over one 6.4-GB buffer, each
// requesting a KNOWN number of bytes (printed), to be run under `rocprofv3 --pmc FETCH_SIZE` (tools/pmc_fetch_calib.sh):
//   stream      every thread reads 16 B, consecutive threads consecutive addresses (the guide's x2 case)
//   rows64      every thread reads one random, 64-B aligned 64-B row as 4 x 16 B (the re-score's code rows; 2^25 rows)
//   dword       every thread reads one random 4-B word (the label gather of knn_interp; 2^25 words)
//   rows128     every thread reads one random, 128-B aligned 128-B row as 8 x 16 B (the star kernel's code rows)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/fetch_calib.hip -o /tmp/fetch_calib && /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
constexpr size_t BYTES = 6400ull << 20;
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__global__ void stream_kernel(const uint4* __restrict__ p, size_t n16, unsigned* out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    const uint4 v = p[i];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) *out = 1;
}
template <int N16, int ALIGN>
__global__ void rows_kernel(const unsigned char* __restrict__ p, size_t n_req, unsigned* out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_req) return;
    const size_t row = mix(i + 1) % (BYTES / ALIGN);
    const uint4* q = reinterpret_cast<const uint4*>(p + row * ALIGN);
    unsigned acc = 0;
#pragma unroll
    for (int e = 0; e < N16; ++e) { const uint4 v = q[e]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) *out = 1;
}
// the same rows fetched COOPERATIVELY: LANES = ALIGN / 16 neighbouring lanes read the 16-B pieces of one row with one instruction
// (one request per row and instruction instead of N16 instructions that each touch 64 different rows)
template <int ALIGN>
__global__ void rows_coop_kernel(const unsigned char* __restrict__ p, size_t n_req, unsigned* out) {
    constexpr int LANES = ALIGN / 16;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;            // thread t: piece t % LANES of rows (t / LANES) + k * (threads / LANES)
    const size_t per_pass = (size_t)gridDim.x * 256 / LANES;
    unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < LANES; ++k) {                                   // LANES independent loads in flight, as in the per-lane kernel
        const size_t i = t / LANES + k * per_pass;
        if (i < n_req) {
            const size_t row = mix(i + 1) % (BYTES / ALIGN);
            const uint4 v = reinterpret_cast<const uint4*>(p + row * ALIGN)[t % LANES];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345u) *out = 1;
}
__global__ void dword_kernel(const unsigned* __restrict__ p, size_t n_req, unsigned* out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_req) return;
    if (p[mix(i + 1) % (BYTES / 4)] == 0x12345u) *out = 1;
}
int main() {
    unsigned char* buf; unsigned* out;
    hipMalloc(&buf, BYTES); hipMalloc(&out, 4);
    hipMemset(buf, 1, BYTES);
    const size_t n_req = 1ull << 25;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(stream_kernel, dim3((unsigned)(BYTES / 16 / 256)), dim3(256), 0, 0, (const uint4*)buf, BYTES / 16, out);
        hipLaunchKernelGGL((rows_kernel<4, 64>), dim3((unsigned)(n_req / 256)), dim3(256), 0, 0, buf, n_req, out);
        hipLaunchKernelGGL(dword_kernel, dim3((unsigned)(n_req / 256)), dim3(256), 0, 0, (const unsigned*)buf, n_req, out);
        hipLaunchKernelGGL((rows_kernel<8, 128>), dim3((unsigned)(n_req / 256)), dim3(256), 0, 0, buf, n_req, out);
        hipLaunchKernelGGL((rows_coop_kernel<64>), dim3((unsigned)(n_req / 256)), dim3(256), 0, 0, buf, n_req, out);
        hipLaunchKernelGGL((rows_coop_kernel<128>), dim3((unsigned)(n_req / 256)), dim3(256), 0, 0, buf, n_req, out);
    }
    hipDeviceSynchronize();
    printf("requested bytes per launch: stream %zu  rows64 %zu  dword %zu  rows128 %zu\n", BYTES, n_req * 64, n_req * 4, n_req * 128);
    return 0;
}
