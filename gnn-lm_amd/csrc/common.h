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

// Opt-in per-kernel timing with HIP events on the launch stream (bench.py's live roofline).
enum KernelId { K_GEMM = 0, K_GATHER = 1, K_STAR = 2, K_CHAIN = 3, K_CAUSAL = 4, K_LAYERNORM = 5, K_LSE = 6,
                K_KNN = 7, K_MISC = 8, K_SPLIT = 9, K_COUNT = 10 };
extern unsigned g_prof_mask;
void prof_start(int kid, hipStream_t s);
// flops / bytes: algorithmic work of the launch; if scale_dev != null the work is multiplied by
// (*scale_dev / scale_den) read back stream-ordered (device-side row counts)
void prof_stop(int kid, hipStream_t s, double flops, double bytes, const int32_t* scale_dev, double scale_den);
struct ProfScope {
    int kid; hipStream_t s; double flops, bytes; const int32_t* sd; double den; bool on;
    ProfScope(int kid_, hipStream_t s_, double flops_, double bytes_, const int32_t* sd_ = nullptr, double den_ = 1.0)
        : kid(kid_), s(s_), flops(flops_), bytes(bytes_), sd(sd_), den(den_), on((g_prof_mask >> kid_) & 1u) {
        if (on) prof_start(kid, s);
    }
    ~ProfScope() { if (on) prof_stop(kid, s, flops, bytes, sd, den); }
};

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
