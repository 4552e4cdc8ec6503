// Shared helpers for the gfx950 kernels of the GNN-LM eval hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

namespace gnnlm {

// Error codes of the C ABI (include/gnnlm.h).
enum { OK = 0, E_INVALID = -22 /*EINVAL*/, E_NOMEM = -12 /*ENOMEM*/, E_HIP = -5 /*EIO*/ };

void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define GNNLM_HIP(call)                                                        \
    do {                                                                       \
        hipError_t _e = (call);                                                \
        if (_e != hipSuccess) return ::gnnlm::hip_fail(_e, #call, __FILE__, __LINE__); \
    } while (0)

#define GNNLM_REQUIRE(cond, msg)                                               \
    do {                                                                       \
        if (!(cond)) {                                                         \
            ::gnnlm::set_error(std::string("invalid argument: ") + msg + " [" #cond "]"); \
            return ::gnnlm::E_INVALID;                                         \
        }                                                                      \
    } while (0)

#define GNNLM_LAUNCH_CHECK() GNNLM_HIP(hipGetLastError())

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// MI355X: 8 XCDs, block b is placed on XCD b % 8 (speed only, never correctness).
// Bijective remap giving every XCD a contiguous chunk of the logical tile list so that
// neighbouring tiles (which share operand panels) hit the same per-XCD L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned nx = 8;
    if (nwg < nx) return bid;
    unsigned xcd = bid % nx, idx = bid / nx;
    unsigned q = nwg / nx, r = nwg % nx;
    unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace gnnlm
