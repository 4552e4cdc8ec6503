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
                K_KNN = 7, K_MISC = 8, K_SPLIT = 9, K_TOPK = 10, K_IVF = 11, K_IVF8 = 12, K_RESCORE = 13, K_IVF8S = 14, K_TAU = 15, K_COUNT = 16 };
extern unsigned g_prof_mask;
void prof_start(int kid, hipStream_t s);
// flops / bytes: algorithmic work of the launch; if scale_dev != null the work is multiplied by
// (*scale_dev / scale_den) read back stream-ordered (device-side row counts)
void prof_stop(int kid, hipStream_t s, double flops, double bytes, const int32_t* scale_dev, double scale_den,
               int32_t* host_slot);
int32_t* prof_take_slot();      // pinned, device-visible int32 for a kernel to store its device-side row count in
struct ProfScope {
    int kid; hipStream_t s; double flops, bytes; const int32_t* sd; double den; bool on;
    int32_t* slot = nullptr;    // kernels_write_count: the launch stores the count here itself (no D2H copy afterwards)
    ProfScope(int kid_, hipStream_t s_, double flops_, double bytes_, const int32_t* sd_ = nullptr, double den_ = 1.0,
              bool kernel_writes_count = false)
        : kid(kid_), s(s_), flops(flops_), bytes(bytes_), sd(sd_), den(den_), on((g_prof_mask >> kid_) & 1u) {
        if (on) {
            if (sd && kernel_writes_count) slot = prof_take_slot();
            prof_start(kid, s);
        }
    }
    ~ProfScope() { if (on) prof_stop(kid, s, flops, bytes, sd, den, slot); }
};

// Dynamic LDS beyond 64 KiB has to be allowed per kernel function AND per device (hipFuncSetAttribute acts on the current
// device's copy of the function): done once per (function, device), whichever thread or device comes first.
int lds_opt_in(const void* kernel_fn, int bytes);
#define GNNLM_LDS_OPT_IN(fn, bytes)                                                          \
    do {                                                                                     \
        const int _rc = ::gnnlm::lds_opt_in(reinterpret_cast<const void*>(fn), (int)(bytes)); \
        if (_rc != ::gnnlm::OK) return _rc;                                                  \
    } while (0)

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

// Cross-lane reductions on the VALU (DPP + the gfx950 permlane swaps): no LDS round trips, unlike
// __shfl_xor (ds_bpermute_b32).  Fixed combination order, every lane ends up with the result.
//   row16_*: over the 16 lanes of a DPP row;  half32_*: over lanes 0..31 / 32..63;  wave_*: all 64.
#define GNNLM_DPP(v_, ctrl_) __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v_), __float_as_int(v_), ctrl_, 0xf, 0xf, false))
__device__ __forceinline__ float row16_sum(float v) {
    v += GNNLM_DPP(v, 0xB1);     // quad_perm [1,0,3,2]
    v += GNNLM_DPP(v, 0x4E);     // quad_perm [2,3,0,1]
    v += GNNLM_DPP(v, 0x141);    // row_half_mirror
    v += GNNLM_DPP(v, 0x140);    // row_mirror
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, GNNLM_DPP(v, 0xB1));
    v = fmaxf(v, GNNLM_DPP(v, 0x4E));
    v = fmaxf(v, GNNLM_DPP(v, 0x141));
    v = fmaxf(v, GNNLM_DPP(v, 0x140));
    return v;
}
#undef GNNLM_DPP
typedef unsigned gnnlm_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float half32_sum(float v) {
    v = row16_sum(v);
    const gnnlm_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);          // (row 0, row 1) / (row 2, row 3)
}
__device__ __forceinline__ float half32_max(float v) {
    v = row16_max(v);
    const gnnlm_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = half32_sum(v);
    const gnnlm_u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float wave_max(float v) {
    v = half32_max(v);
    const gnnlm_u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
}

}  // namespace gnnlm
