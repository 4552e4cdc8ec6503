// Internal launcher declarations (C++).  The public C ABI is include/gnnlm.h.
#pragma once
#include "common.h"
#include "../../include/gnnlm.h"

namespace gnnlm {

// Descriptor structs are the public C structs of include/gnnlm.h.
typedef gnnlm_gemm_t GemmParams;
typedef gnnlm_gather_t GatherParams;
typedef gnnlm_star_attn_t StarAttnParams;
typedef gnnlm_chain_attn_t ChainAttnParams;
typedef gnnlm_knn_interp_t KnnInterpParams;
int topk_merge(const gnnlm_topk_t& d, hipStream_t stream);
int ivfpq_scan(const gnnlm_ivfpq_scan_t& d, hipStream_t stream);
int ivfpq_pack_codes(const uint8_t* codes, int64_t N, int M, uint8_t* out, hipStream_t stream);
int ivfpq_pack_lut(const float* lut, int64_t ld_lut, int64_t n, int M, float* out, hipStream_t stream);

int ivfpq_pack_tiles(const uint8_t* codes, int64_t N, int M, uint8_t* out, hipStream_t stream);
int ivfpq_build_groups(const int64_t* pl, int64_t ld, int64_t n, int P, int nlist, int64_t seg, int32_t* grp_list, int32_t* grp_q,
                       int64_t* grp_out, int32_t* n_groups, int32_t* scratch, hipStream_t stream);
int ivfpq_quantize_lut(const float* lut, int64_t ld_lut, int64_t n, int M, uint8_t* qlut, float* qmeta, hipStream_t stream);
int ivfpq_scan8(const gnnlm_ivfpq_scan8_t& d, hipStream_t stream);
int ivfpq_rescore(const gnnlm_ivfpq_rescore_t& d, hipStream_t stream);
int ivfpq_refine(const gnnlm_ivfpq_refine_t& d, hipStream_t stream);
int ivfpq_split_payload(int64_t* idx, int64_t n, int label_bits, int32_t val_last, int32_t* out_vals, hipStream_t stream);
int ivfpq_tau(const gnnlm_ivfpq_tau_t& d, hipStream_t stream);

int gemm_nt(const GemmParams& p, hipStream_t stream);
// big-tile split-bf16 path (gemm_split.hip): taken by gemm_nt for precision != 0 when the problem fills the chip
bool gemm_split_eligible(const GemmParams& p);
int gemm_nt_split(const GemmParams& p, hipStream_t stream);
// f32 MFMA kernel with LDS-DMA staging (gemm_f32_dma.hip): taken by gemm_nt for 128x128-tile problems with K % 16 == 0
bool gemm_dma_eligible(const GemmParams& p);
int gemm_nt_dma(const GemmParams& p, hipStream_t stream);
bool gemm_sched_eligible(const GemmParams& p);
int gemm_nt_sched(const GemmParams& p, hipStream_t stream);
bool gemm_skinny_eligible(const GemmParams& p);
int gemm_nt_skinny(const GemmParams& p, hipStream_t stream);
// precision used by gemm_nt for descriptors that leave `precision` at 0 (set by the orchestrators)
extern thread_local int g_default_gemm_precision;
struct GemmPrecisionScope {
    int saved;
    explicit GemmPrecisionScope(int prec) : saved(g_default_gemm_precision) { g_default_gemm_precision = prec; }
    ~GemmPrecisionScope() { g_default_gemm_precision = saved; }
};
int lse_reduce(const float* part, int n_parts, int64_t rows, const int32_t* m_dev, float* lse, hipStream_t stream);
int gather_decode(const GatherParams& p, hipStream_t stream);
int pq_encode(const float* x, int64_t ldx, const float* cen, const float* norm2, int M, int dsub, int64_t n, uint8_t* codes,
              hipStream_t stream);
int bucket_rows(const int64_t* rows, int64_t n, int64_t n_store, int64_t per, int world, int self, int64_t* counts,
                int64_t* cursor, int64_t* send_rows, int32_t* inv, hipStream_t stream);
int gather_rows_peer(const gnnlm_peer_gather_t& d, hipStream_t stream);
int bucket_rows_padded(const int64_t* rows, int64_t n, int64_t n_store, int64_t per, int world, int64_t cap, int64_t* cursor,
                       int64_t* send_rows, int32_t* inv, int64_t* overflow, hipStream_t stream);
int star_attn(const StarAttnParams& p, hipStream_t stream);
// table-resident formulation for the PQ source, k_g <= 128 (star_tab.hip); star_attn dispatches to it
bool star_attn_tab_eligible(const StarAttnParams& p);
int star_attn_tab(const StarAttnParams& p, hipStream_t stream);

// Neighbour (i, j) takes part in the star softmax iff its id is a row of the store, the row is present on this
// shard (PQ source read from the store) and the caller's validity byte (exchange / gather_decode) says so: the
// rule of gather_decode_kernel, so layer-0 star attention and the ntgt states can never disagree.
// row `row` (>= 0) of a set of mapped shards: pointer to its M bytes, or nullptr if no shard holds it
__device__ __forceinline__ const uint8_t* shard_row_ptr(const gnnlm_shards_t* sh, int64_t row, int M) {
    const int64_t per = sh->rows_per_rank;
    // (a 64-bit division costs ~10x a 32-bit one; the stores of this path have fewer than 2^32 rows)
    const int64_t q = ((uint64_t)(row | per) >> 32) == 0 ? (int64_t)((uint32_t)row / (uint32_t)per) : row / per;
    const int g = (int)min((int64_t)sh->n - 1, q);
    const int64_t local = row - sh->row0[g];
    return local >= 0 && local < sh->rows[g] ? sh->base[g] + local * M : nullptr;
}
// group of neighbour (i, j): its own (i * kg + j), or the distinct-centre group of a de-duplicated batch (-1: none)
__device__ __forceinline__ int64_t star_group(const StarAttnParams& p, int i, int j) {
    const int64_t e = (int64_t)i * p.kg + j;
    return p.x_index ? (int64_t)p.x_index[e] : e;
}
__device__ __forceinline__ bool star_nb_ok(const StarAttnParams& p, int i, int j, int64_t id) {
    bool ok = id >= 0 && (p.n_store <= 0 || id < p.n_store);
    if (p.shards) ok = ok && shard_row_ptr(p.shards, id, p.M) != nullptr;
    else if (p.codes && !p.codes_direct) ok = ok && id - p.row0 >= 0 && id - p.row0 < p.n_local;
    if (ok && (p.nb_valid || p.x_index)) {
        const int64_t gi = star_group(p, i, j);
        ok = gi >= 0 && (!p.nb_valid || p.nb_valid[gi * p.nb_valid_stride] != 0);
    }
    return ok;
}
// row of the code table that holds neighbour (i, j) (store rows, or the slots of an exchange)
__device__ __forceinline__ int64_t star_code_row(const StarAttnParams& p, int i, int j, int64_t id) {
    if (!p.codes_direct) return id - p.row0;
    const int64_t s = ((int64_t)i * p.kg + j) * p.codes_direct;
    return p.codes_index ? (int64_t)p.codes_index[s] : s;
}
// pointer to the M code bytes of a VALID neighbour (i, j) (star_nb_ok): one table, the slots of an exchange, or mapped shards
__device__ __forceinline__ const uint8_t* star_code_ptr(const StarAttnParams& p, int i, int j, int64_t id) {
    if (p.shards) return shard_row_ptr(p.shards, id, p.M);
    return p.codes + star_code_row(p, i, j, id) * p.M;
}
// dense neighbour rows (layers >= 1), D in {256, 512, 1024}, H <= 8: one pass over the rows with a running softmax (star_dense.hip)
bool star_attn_dense_eligible(const StarAttnParams& p);
int star_attn_dense(const StarAttnParams& p, hipStream_t stream);
int chain_attn(const ChainAttnParams& p, hipStream_t stream);

// rows of S[b, h, w, :T] -> causal softmax (u <= w, and w-u < max_ctx if max_ctx > 0), in place
int causal_softmax(float* S, int64_t n_mats, int T, int64_t ld, int max_ctx, hipStream_t stream);
// fused scores + masked softmax + P.V for T = 256, d_k = 128 (attn.hip); Q, K', V are [n_blocks * T, ld] with head h at
// column h * dk
bool causal_attn_fused_ok(int T, int dk);
int causal_attn_fused(const float* Q, const float* K, const float* V, int64_t ld, float* out, int64_t ldo,
                      int n_blocks, int T, int H, int dk, int max_ctx, hipStream_t stream, bool accumulate = false);

// out[r,:] = LayerNorm(x[r,:]) * gamma + beta ; optional row validity (invalid rows -> 0)
int layernorm(const float* x, int64_t ldx, const float* gamma, const float* beta, float* out, int64_t ldo,
              int64_t rows, int d, float eps, const uint8_t* valid, hipStream_t stream,
              const float* residual = nullptr, int64_t ldr = 0,      // out = LN(x + residual)
              const int32_t* rows_idx = nullptr,                     // only the rows rows_idx[0..rows) (of x, residual, out, valid)
              const int32_t* n_dev = nullptr, int n_mult = 1);       // device-side count: rows = min(rows, *n_dev * n_mult)
// idx[g * n_sel + j] = g * n_g + sel[j]
int group_rows(int32_t* idx, int64_t n_groups, int n_g, const int* sel, int n_sel, hipStream_t stream);
// dst[slots[g]][0..d) = src[g * ld_src][0..d) for g < n (rows of d floats, d % 4 == 0)
int scatter_rows(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, const int32_t* slots, int64_t n, int d, hipStream_t stream,
                 const int32_t* n_dev = nullptr);                    // device-side count: n = min(n, *n_dev)

// group assignment on the device (groups.hip, ABI 9)
int group_assign(const gnnlm_group_assign_t& d, hipStream_t stream);
// (ABI 11: row-keyed K / V of layer 0) slot -> datastore row of the first *n_dev groups, for the slots within `reach` of their centre
int slot_rows(const int64_t* centres, int64_t G, const int32_t* n_dev, int left, int right, int reach, int64_t n_store, int64_t* out, hipStream_t stream);
// counts[w * 8 + mult - 1] = rows of window w (groups [w * per, (w + 1) * per) of min(*n_dev, cap)) times mult, mult = 1 .. 8
int window_counts(const int32_t* n_dev, int64_t cap, int64_t per, int n_win, int32_t* counts, hipStream_t stream);
// out[e] = row of the fetched buffer that holds the centre code of neighbour e's group (-1: none)
int nb_code_rows(const int32_t* group_index, const int32_t* fetched_index, int n_g, int64_t n, int32_t* out, hipStream_t stream);
// dst[slots[g]] = src[index ? index[g * stride] : g * stride] (zero row if valid && !valid[g * stride]) for g < min(*n_dev, cap)
int scatter_code_rows(const uint8_t* src, const int32_t* index, int64_t stride, const uint8_t* valid, uint8_t* dst, const int32_t* slots,
                      const int32_t* n_dev, int64_t cap, int bytes, hipStream_t stream);

// out = 0.5 * (a + b)
int mean2(const float* a, const float* b, float* out, int64_t n, hipStream_t stream);

int gelu(float* x, int64_t n, hipStream_t stream);

// the --invalid-neighbor-context rule of token_block_dataset.py:360-362 applied to a batch of neighbour-id rows
int filter_neighbors(const int64_t* ids, const int64_t* pos, int64_t n, int kg, int64_t ctx, int64_t* out, hipStream_t stream);

// fp16 -> fp32 row convert
int half_to_float(const void* src, float* dst, int64_t n, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// adaptive softmax / kNN interpolation
// ---------------------------------------------------------------------------------------------
// lse[r] = logsumexp(logits[r, :n]); picked[r] = logits[r, pick[r]] (pick may be null)
int row_lse_pick(const float* logits, int64_t ld, int64_t rows, const int32_t* m_dev, int n,
                 const int32_t* pick, float* lse, float* picked, hipStream_t stream);

struct BandSplitParams {
    const int64_t* target = nullptr; int64_t n = 0;
    int n_bands = 0; int32_t cutoff[8] = {0};      // cutoff[0..n_bands-1], band b = [cutoff[b-1], cutoff[b])
    int32_t* head_pick = nullptr;          // [n] index into the head logits (target or cutoff0+band-1)
    int32_t* band_rows = nullptr;          // [n_bands-1, n] compacted row lists of the tail bands
    int32_t* band_pick = nullptr;          // [n_bands-1, n] target - cutoff[band-1] for the compacted rows
    int32_t* band_count = nullptr;         // [n_bands-1]
};
int band_split(const BandSplitParams& p, hipStream_t stream);

// lm_logp[rows[r]] = head_lsm[rows[r]] + (tail_picked[r] - tail_lse[r])   (rows null: identity)
int head_logp(const float* picked, const float* lse, float* out, int64_t n, hipStream_t stream);
int tail_combine(const float* tail_picked, const float* tail_lse, const int32_t* rows,
                 const int32_t* count_dev, int64_t n_max, float* lm_logp, hipStream_t stream);

int knn_interp(const KnnInterpParams& p, hipStream_t stream);
int label_tags(const void* vals, int itemsize, int64_t n, uint8_t* tag, hipStream_t stream);
size_t knn_interp_scratch_bytes(int64_t n, int k, int64_t n_local);
bool knn_interp_bucketed_eligible(const KnnInterpParams& p);
int knn_interp_bucketed(const KnnInterpParams& p, float log_1ml, float log_l, hipStream_t stream);

// sum of x[start[b] : ] per ... simple masked sum in double: out[0] += sum(x[i] * (mask?mask[i]:1))
int masked_sum_f64(const float* x, const uint8_t* mask, int64_t n, double* out, hipStream_t stream);

}  // namespace gnnlm
