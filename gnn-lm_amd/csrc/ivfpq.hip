// IVF-PQ list scan with LDS-resident ADC look-up tables (see gnnlm_ivfpq_scan_t in include/gnnlm.h): the scoring half
// of the on-device replacement of the reference's faiss CPU search (knn/knn_model.py:87-101; index
// `OPQ64_1024,IVF4096,PQ64`, nprobe 32: gnnlm_scripts/wiki103/find_knn.sh:8-13).
//
// One workgroup per (query, probed list) task.  The query's table (M x 256 floats: 64 KiB at M = 64) is copied to
// LDS once; every thread then scores keys of the list: its code row (M bytes, 16-B loads) and M table look-ups.
// Tasks arrive grouped by list, so the workgroups running at the same time read the same code rows out of L2 / the
// Infinity Cache; a list is read from HBM about once per search batch.  Bound: the LDS look-up rate (M random 4-B
// reads per (query, key) pair).
#include "kernels.h"

namespace gnnlm {
namespace {

__global__ __launch_bounds__(256) void ivfpq_scan_kernel(gnnlm_ivfpq_scan_t p) {
    extern __shared__ __attribute__((aligned(16))) float lut[];            // [M][256]
    const int tid = threadIdx.x;
    const int64_t task = blockIdx.x;
    const int q = p.task_q[task], slot = p.task_p[task];
    const int64_t list = p.probe_list[(int64_t)q * p.ld_probe + slot];
    const int M = p.M;
    int64_t lo = 0, hi = 0;
    if (list >= 0) { lo = p.list_off[list]; hi = p.list_off[list + 1]; }
    const int64_t len = hi - lo;
    float* oval = nullptr;
    int64_t* oid = nullptr;
    if (!p.tau) {
        oval = p.out_val + (int64_t)q * p.ld_out + (int64_t)(slot - p.p0) * p.seg;
        oid = p.out_id + (int64_t)q * p.ld_out + (int64_t)(slot - p.p0) * p.seg;
        for (int64_t j = min(len, (int64_t)p.seg) + tid; j < p.seg; j += 256) oid[j] = -1;     // beyond the list
    }
    if (len == 0) return;
    {   // the query's table -> LDS (coalesced 16-B pieces)
        const float4* src = reinterpret_cast<const float4*>(p.lut + (int64_t)q * p.ld_lut);
        float4* dst = reinterpret_cast<float4*>(lut);
        for (int e = tid; e < M * 64; e += 256) dst[e] = src[e];
    }
    __syncthreads();
    const float bias = p.probe_bias[(int64_t)q * p.ld_probe + slot];
    const float tau = p.tau ? p.tau[q] : 0.f;
    for (int64_t j = tid; j < len; j += 256) {
        const uint4* crow = reinterpret_cast<const uint4*>(p.codes + (lo + j) * M);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;                       // four chains: the look-ups are independent
        for (int c16 = 0; c16 < M / 16; ++c16) {
            const uint4 v = crow[c16];
            const float* t = lut + c16 * 16 * 256;
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s0 += t[(4 * u + 0) * 256 + (w[u] & 255u)];
                s1 += t[(4 * u + 1) * 256 + ((w[u] >> 8) & 255u)];
                s2 += t[(4 * u + 2) * 256 + ((w[u] >> 16) & 255u)];
                s3 += t[(4 * u + 3) * 256 + (w[u] >> 24)];
            }
        }
        const float s = bias + ((s0 + s1) + (s2 + s3));
        if (!p.tau) {
            if (j < p.seg) { oval[j] = s; oid[j] = p.ids[lo + j]; }
        } else if (s > tau) {
            const int pos = atomicAdd(&p.cand_cnt[q], 1);
            if (pos < p.cap) {
                p.cand_val[(int64_t)q * p.cap + pos] = s;
                p.cand_id[(int64_t)q * p.cap + pos] = p.ids[lo + j];
            }
        }
    }
}

}  // namespace

int ivfpq_scan(const gnnlm_ivfpq_scan_t& d, hipStream_t stream) {
    GNNLM_REQUIRE(d.n_tasks >= 0 && d.n_tasks < (1ll << 31), "ivfpq_scan: bad task count");
    if (d.n_tasks == 0) return OK;
    GNNLM_REQUIRE(d.codes && d.ids && d.list_off && d.lut && d.probe_list && d.probe_bias && d.task_q && d.task_p,
                  "ivfpq_scan: null operand");
    GNNLM_REQUIRE(d.M > 0 && d.M % 16 == 0 && d.M <= 128 && d.ld_lut >= (int64_t)d.M * 256 && d.ld_lut % 4 == 0 &&
                      (uintptr_t)d.lut % 16 == 0 && (uintptr_t)d.codes % 16 == 0,
                  "ivfpq_scan: need M % 16 == 0, M <= 128, 16-byte aligned tables");
    if (d.tau) GNNLM_REQUIRE(d.cand_val && d.cand_id && d.cand_cnt && d.cap > 0, "ivfpq_scan: filtered mode needs the candidate buffers");
    else GNNLM_REQUIRE(d.out_val && d.out_id && d.seg > 0 && d.ld_out >= d.seg, "ivfpq_scan: dense mode needs the output rows");
    const size_t lds = (size_t)d.M * 256 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        GNNLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ivfpq_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_set = true;
    }
    ProfScope prof(K_IVF, stream, 0.0, 0.0);
    hipLaunchKernelGGL(ivfpq_scan_kernel, dim3((unsigned)d.n_tasks), dim3(256), lds, stream, d);
    GNNLM_LAUNCH_CHECK();
    return OK;
}

}  // namespace gnnlm
