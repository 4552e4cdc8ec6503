/* gnnlm.h -- C ABI of libgnnlm_hip.so: the MI355X (gfx950) implementation of the GNN+kNN eval hot
 * path of ShannonAI/GNN-LM (`fairseq-eval-lm --graph --use-precompute-feat --knnlm`).
 *
 * The reference has NO native / FFI boundary on this path (everything is Python over torch, DGL and
 * faiss: SURVEY.md section 8b), so each entry point below names the reference Python function it
 * replaces (paths relative to the reference tree).  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative errno-style code (GNNLM_E_*);
 *     gnnlm_last_error() returns a thread-local message for the last failure
 *   - all `const float*` / `void*` operands are DEVICE pointers unless the name says `host`
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls only enqueue work
 *   - the caller owns every buffer; descriptors are plain structs, zero-initialise them and fill
 *     what you need (0 / NULL always means "not used / default")
 *   - matrices are row-major; "ld" is the row stride in ELEMENTS
 *   - no call allocates or synchronises except gnnlm_store_* (explicitly documented)
 */
#ifndef GNNLM_H
#define GNNLM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNNLM_ABI_VERSION 11
#define GNNLM_OK 0
#define GNNLM_E_INVALID (-22)
#define GNNLM_E_NOMEM (-12)
#define GNNLM_E_HIP (-5)

const char* gnnlm_last_error(void);
int gnnlm_abi_version(void);
/* name of the code object's target ("gfx950") -- lets a loader check it got the MI355X build */
const char* gnnlm_target_arch(void);
/* sizeof() of a descriptor struct by its typedef name ("gnnlm_gemm_t", ...), 0 if unknown: lets a
 * foreign-language binding verify its struct mirror */
size_t gnnlm_sizeof(const char* struct_name);

/* ------------------------------------------------------------------------------------------------
 * Dense contraction:  C[M,N] = alpha * A[M,K] . W[N,K]^T (+ gate[row] * bias) (+ R)
 * f32 operands, f32 MFMA accumulate (v_mfma_f32_32x32x2_f32: exact fmaf chain).
 * Replaces: nn.Linear of HGTLayer (fairseq/models/hgt.py:315-322,401), the relation einsum (:347-348),
 *           fn.v_dot_u / u_mul_e+sum on the dense tgt-tgt edges (:354,383-385),
 *           `x @ A` of TorchPQCodec.decode (knn/pq_wrapper.py:202),
 *           the head/tail matmuls of AdaptiveSoftmax (fairseq/modules/adaptive_softmax.py:184-203).
 * ---------------------------------------------------------------------------------------------- */
typedef struct gnnlm_gemm {
    const float* A;  int64_t lda;
    const int32_t* a_rows;     /* optional gather: logical row r reads A row a_rows[r] (<0: zero row) */
    const float* W;  int64_t ldw;
    float* C;        int64_t ldc;
    const int32_t* c_rows;     /* optional scatter: logical row r is stored to C (and read from R) row c_rows[r] */
    const float* bias;         /* bias_mode 1: per column [N]; 2: per row [M] */
    int32_t bias_mode;
    const float* gate;         /* optional [M]: per-row multiplier of the bias */
    const float* R;  int64_t ldr;
    float alpha;               /* 0 is read as 1 */
    int32_t M, N, K;           /* K % 4 == 0 */
    const int32_t* m_dev;      /* optional device-side row count (<= M): tiles beyond it exit */
    int32_t* m_out;            /* optional: receives min(M, *m_dev) (device or host-mapped memory; used by the profiler) */
    int32_t batch1, batch2;    /* 0 is read as 1; batch index (b1, b2) */
    int64_t sA1, sA2, sW1, sW2, sC1, sC2, sB1, sB2, sR1, sR2;   /* batch strides in elements */
    int32_t precision;         /* 0: f32 MFMA (exact fmaf chain; or the enclosing orchestrator's setting);
                                  1: bf16x3 split emulation (~2^-16 per product); 2: bf16x6 (~2^-24, f32-level) */
    int32_t tile_order;        /* 0: auto (consecutive tiles share the larger operand's panel), 1: n fastest, 2: m fastest */
    /* log-sum-exp epilogue (C may be NULL): instead of storing C, every (row, 64-column slab) writes a
     * (max, sum exp(x - max)) pair to lse_part[row][slab], slab count = 2*ceil(N/128); lse_picked[row] =
     * alpha * (A.W^T)[row, lse_pick[row]].  Finish with gnnlm_lse_reduce.  batch must be 1. */
    float* lse_part;           /* [M, 2*ceil(N/128), 2] */
    const int32_t* lse_pick;   /* optional [M] */
    float* lse_picked;         /* [M] (with lse_pick) */
    int64_t a_rows_bound;      /* ABI 5, optional with a_rows: every a_rows[r] < a_rows_bound (0: unknown) -- lets the kernels with
                                  32-bit row offsets take a gathered problem */
} gnnlm_gemm_t;
int gnnlm_gemm_nt(const gnnlm_gemm_t* desc, void* stream);
/* lse[row] = log sum exp over the row, from the partial pairs of the LSE epilogue */
int gnnlm_lse_reduce(const float* part, int32_t n_parts, int64_t rows, const int32_t* m_dev, float* lse, void* stream);

/* ------------------------------------------------------------------------------------------------
 * PQ datastore row gather + decode (HBM-resident store).
 * Replaces: quant_neighbor_feats[offset] / neighbor_tokens[offset] row gathers and the context
 *           expansion of GraphTokenBlockDataset.new_build_graph
 *           (fairseq/data/token_block_dataset.py:354-394,407-410; PlasmaArray.__getitem__
 *           fairseq/data/new_plasma_utils.py:115-116; MmapDataset.__getitem__ fairseq/data/mmap_dataset.py:57-58)
 *           and the lookup half of TorchPQCodec.decode (knn/pq_wrapper.py:169-196).
 * Slot order inside a group: centre o, then o-left..o-1, then o+1..o+right (reference node order).
 * A slot is valid iff ids[g] != -1 and 0 <= row < n_store (the bound of :384) and the
 * row lies in the shard [row0, row0+n_local).
 * ---------------------------------------------------------------------------------------------- */
/* A range-sharded code table whose shards are ALL mapped into this process (the local one, the peers' through HIP IPC:
 * gnnlm_amd.dist.PeerMappedFetcher): the kernels that read code rows then take row r from shard min(n - 1, r /
 * rows_per_rank) -- over xGMI straight from the owner's HBM, no exchange and no copy.  The table lives in DEVICE memory;
 * the descriptors carry a pointer to it (NULL: not used, codes / row0 / n_local describe one table) -- by value it would
 * add 400 B of kernel arguments to every launch.  base[g] holds the rows [row0[g], row0[g] + rows[g]) (rank g's range plus its halo). */
typedef struct gnnlm_shards {
    int32_t n, reserved;
    int64_t rows_per_rank;
    const uint8_t* base[16];
    int64_t row0[16], rows[16];
} gnnlm_shards_t;

typedef struct gnnlm_gather {
    const uint8_t* codes;      /* [n_local, M] */
    const void* vals;          /* [n_local] int16 / int32 (optional) */
    int32_t vals_itemsize;     /* 2 or 4 */
    int64_t n_store, row0, n_local;
    int32_t M, dsub;           /* dsub % 4 == 0; ksub is fixed to 256 (8-bit codes, pq_wrapper.py:33) */
    const float* centroids;    /* [M, 256, dsub] */
    const int64_t* ids;        /* [n_groups] */
    int64_t n_groups;
    int32_t left, right;
    float* out_x;  int64_t ld_x;   /* optional [n_groups*(1+left+right), M*dsub], zero rows for invalid slots */
    uint8_t* out_codes;        /* optional [n_slots, M] */
    int32_t* out_labels;       /* optional [n_slots], -1 for invalid */
    uint8_t* out_valid;        /* optional [n_slots] */
    int32_t direct;            /* 1: `codes` is an already-fetched [n_slots, M] buffer (slot s = row s), */
    const uint8_t* in_valid;   /*    validity comes from in_valid[n_slots]; ids is ignored */
    const int32_t* in_index;   /*    optional with direct: slot s reads row in_index[s] of `codes` */
    const gnnlm_shards_t* shards;  /* ABI 4 (DEVICE pointer): replaces codes / row0 / n_local for the code rows (not with direct, not for vals) */
    const int32_t* n_groups_dev;   /* ABI 9, optional (DEVICE int32): only the first min(n_groups, *n_groups_dev) groups are processed */
} gnnlm_gather_t;
int gnnlm_pq_gather_decode(const gnnlm_gather_t* desc, void* stream);

/* PQ encode of already-rotated rows: codes[r, m] = argmin_c norm2[m, c] - 2 x[r, m*dsub:(m+1)*dsub] . centroids[m, c]
 * (lowest index on ties).  Replaces the argmin of TorchPQCodec.encode (knn/pq_wrapper.py:131-167); the OPQ
 * rotation `x @ A.T + b` in front of it is a gnnlm_gemm_nt.  Offline producer in the reference
 * (knn/quantize_features.py:122-146), a "next" row of the hot-path table. */
int gnnlm_pq_encode(const float* x, int64_t ldx, const float* centroids, const float* norm2, int32_t M, int32_t dsub,
                    int64_t n, uint8_t* codes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ('ntgt','inter','tgt') attention with the neighbour-side projections absorbed into the query:
 *   s[i,h,j] = x_j . U[i,h,:]  (valid j only),  alpha = softmax_j,  Z[i,h,:] = sum_j alpha x_j
 * x_j is decoded on the fly from PQ codes (codes != NULL) or read from dense rows (X != NULL).
 * Replaces: apply_edges(v_dot_u) + edge_softmax + u_mul_e/sum on the star edges
 *           (fairseq/models/hgt.py:354-356,383-385) fused with the gather/decode above.
 * ---------------------------------------------------------------------------------------------- */
typedef struct gnnlm_star_attn {
    const float* U;            /* [T, H, D] */
    const int64_t* ids;        /* [T, kg], -1 = no neighbour */
    int32_t T, H, D, kg;
    const uint8_t* codes;  int64_t row0, n_local;  int32_t M, dsub;
    int32_t codes_direct;      /* s > 0: `codes` is an already-fetched buffer, row of (i,j) = (i*kg+j)*s (sharded store) */
    const float* centroids;
    const float* X;  int64_t ldx;  int64_t x_group_stride;   /* dense row of (i,j) = X + (i*kg+j)*x_group_stride*ldx */
    float* Z;                  /* [T, H, D] */
    float* has_nb;             /* optional [T]: 1.0 if >= 1 valid neighbour */
    const int32_t* codes_index;   /* optional with codes_direct: row of (i,j) = codes_index[(i*kg+j)*codes_direct] */
    /* Neighbour validity (ABI 2).  (i,j) takes part in the softmax iff ids[i,j] >= 0, ids[i,j] < n_store (when
     * n_store > 0), the row lies in the shard [row0, row0+n_local) (PQ source read from the store itself) and
     * nb_valid[(i*kg+j)*nb_valid_stride] != 0 (when nb_valid is given: the validity bytes of an exchange / of
     * gnnlm_pq_gather_decode) -- the same rule as gnnlm_pq_gather_decode, so the star edges and the ntgt states
     * always agree.  The reference raises IndexError for rows >= n_store (token_block_dataset.py:370). */
    int64_t n_store;
    const uint8_t* nb_valid;  int64_t nb_valid_stride;
    const gnnlm_shards_t* shards;  /* ABI 4 (DEVICE pointer): replaces codes / row0 / n_local (PQ source, not with codes_direct) */
    /* ABI 6: optional [T, kg] group of neighbour (i, j) (-1: not a neighbour) when the dense rows / validity bytes are stored
     * once per DISTINCT centre row of the batch: row of (i,j) = X + x_index[i*kg+j]*x_group_stride*ldx, validity byte
     * nb_valid[x_index[i*kg+j]*nb_valid_stride] */
    const int32_t* x_index;
} gnnlm_star_attn_t;
int gnnlm_star_attn(const gnnlm_star_attn_t* desc, void* stream);

/* ('ntgt','intra','ntgt'): path graph with self loops over each group's slots
 * (build_ntgt_edges(context=1, bidirect=True), fairseq/data/token_block_dataset.py:395-398,545-584;
 *  attention math fairseq/models/hgt.py:354-356,383-385). */
typedef struct gnnlm_chain_attn {
    const float* Q;  const float* K;  const float* V;  int64_t ld;   /* [n_slots, d] */
    const uint8_t* valid;      /* [n_slots] */
    int64_t n_groups;  int32_t left, right, H, dk;
    const float* scale;        /* optional [H]; NULL = 1 (relation_pri/sqrt(dk) folded into K) */
    float* out;  int64_t ldo;
    int32_t radius_p1;         /* ABI 4.  0: every slot is computed; r + 1 > 0: only the slots within r positions of the centre
                                  are (Q is read for them, K / V up to r + 1 positions away; the other rows of `out` stay untouched) */
    const int32_t* n_groups_dev;   /* ABI 9, optional (DEVICE int32): only the first min(n_groups, *n_groups_dev) groups */
    const int32_t* kv_index;       /* ABI 11, optional [n_slots]: slot s reads row kv_index[s] of K and V (Q, valid and out stay slot-indexed) */
} gnnlm_chain_attn_t;
int gnnlm_chain_attn(const gnnlm_chain_attn_t* desc, void* stream);

/* masked row softmax of the dense ('tgt','intra','tgt') scores (auto_regressive_edges,
 * fairseq/data/token_block_dataset.py:586-594; edge_softmax fairseq/models/hgt.py:356).
 * S holds n_mats matrices of T rows, row stride ld; columns >= T are zeroed. */
int gnnlm_causal_softmax(float* S, int64_t n_mats, int32_t T, int64_t ld, int32_t max_ctx, void* stream);
/* The three steps above in one kernel for the recipe shape T = 256, d_k = 128 (other shapes: -EINVAL, use the GEMM +
 * gnnlm_causal_softmax + GEMM sequence): Q, K', V' are [n_blocks*T, ld] f32 with head h at columns [h*dk, (h+1)*dk);
 * out[blk*T + w, h*dk + :] = sum_{u <= w, w-u < max_ctx} softmax_u(Q_w . K'_u) V'_u.  The score matrix stays in
 * registers.  Replaces fn.v_dot_u + edge_softmax + u_mul_e/sum on ('tgt','intra','tgt') (hgt.py:354-356,383-385). */
int gnnlm_causal_attn(const float* Q, const float* K, const float* V, int64_t ld, float* out, int64_t ldo,
                      int32_t n_blocks, int32_t T, int32_t H, int32_t dk, int32_t max_ctx, void* stream);

/* LayerNorm epilogue of HGTLayer (fairseq/models/hgt.py:404-405).  valid (optional): rows with 0 -> zeros */
int gnnlm_layernorm(const float* x, int64_t ldx, const float* gamma, const float* beta, float* out,
                    int64_t ldo, int64_t rows, int32_t d, float eps, const uint8_t* valid, void* stream);

/* `--invalid-neighbor-context c` (fairseq/data/token_block_dataset.py:360-362, switched on for the train split only:
 * fairseq/tasks/language_modeling.py:299): a neighbour whose datastore row lies within c positions of its own token
 * (`abs(offsets[tgt_idx] - offset) < c`) is skipped exactly like a -1 id -- so the rule is an id rewrite in front of the
 * graph consumers: out[i, j] = -1 if ids[i, j] != -1 and |tok_pos[i] - ids[i, j]| < c, else ids[i, j].
 * ids / out [n_tok, kg] int64 (out may alias ids), tok_pos [n_tok] int64 = the tokens' global offsets in the split. */
int gnnlm_filter_neighbors(const int64_t* ids, const int64_t* tok_pos, int64_t n_tok, int32_t kg, int64_t invalid_ctx,
                           int64_t* out, void* stream);

/* fp16 -> fp32 (precompute_feats[offsets].astype(np.float32), token_block_dataset.py:328) */
int gnnlm_half_to_float(const void* src, float* dst, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Tied adaptive softmax, target log-probability only
 * (AdaptiveSoftmax.get_log_prob with target, fairseq/modules/adaptive_softmax.py:170-206, +
 *  gather_target_probs fairseq/sequence_scorer.py:48-53,89).
 * ---------------------------------------------------------------------------------------------- */
int gnnlm_row_lse_pick(const float* logits, int64_t ld, int64_t rows, const int32_t* m_dev, int32_t n,
                       const int32_t* pick, float* lse, float* picked, void* stream);

typedef struct gnnlm_adaptive_softmax {
    int32_t d, n_bands;
    int32_t cutoff[8];         /* cutoff[0..n_bands-1]; cutoff[n_bands-1] = vocab size */
    int32_t gemm_precision;    /* precision of the GEMMs (see gnnlm_gemm_t.precision); 0 = exact f32 */
    const float* head_w;       /* [cutoff[0] + n_bands - 1, d]: rows of E_0 followed by class_proj */
    const float* proj_t[8];    /* band b>=1: [dim_b, d] (embeddings.b.1.weight TRANSPOSED) */
    const float* emb[8];       /* band b>=1: [cutoff[b]-cutoff[b-1], dim_b] */
    int32_t dim[8];
} gnnlm_adaptive_softmax_t;
size_t gnnlm_adaptive_workspace_bytes(const gnnlm_adaptive_softmax_t* w, int64_t n);
/* lm_logp[r] = log p(target[r] | x[r]) */
int gnnlm_adaptive_target_logp(const gnnlm_adaptive_softmax_t* w, const float* x, int64_t ldx,
                               const int64_t* target, int64_t n, float* lm_logp,
                               void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * kNN-LM distance-softmax + interpolation
 * (KNNModel.get_knn_prob knn/knn_model.py:192-217; SequenceScorer fairseq/sequence_scorer.py:55-68,110,121).
 * ---------------------------------------------------------------------------------------------- */
typedef struct gnnlm_knn_interp {
    const float* lm_logp;      /* [n] */
    const float* sims;         /* [n, k] similarities after sim_func (knn_model.py:137-177) */
    const int64_t* ids;        /* [n, k], -1 = padding (masked with -1e10, :193) */
    const void* vals;  int32_t vals_itemsize;  int64_t n_store, row0, n_local;
    const int32_t* knn_vals;   /* optional [n, k]: vals[ids] already fetched (sharded store) */
    const int64_t* targets;    /* [n] */
    int64_t n;  int32_t k;
    float temperature;
    double lmbda;
    float* out_logp;           /* [n] */
    float* out_pknn;           /* optional [n] */
    int64_t* out_recall;       /* optional [n] */
    /* ABI 7, optional: one TAG byte per label row, vals_tag[r] = gnnlm_label_tag(vals[r]) (built once per store by
     * gnnlm_label_tags).  A neighbour can only match the target when the tags match, so the k gathers of `vals[knns]`
     * (knn_model.py:198) read this 4x smaller table and the 4-byte label only on a tag match (1 in 256 + the true hits):
     * same result bit for bit, a quarter of the table behind the random reads. */
    const uint8_t* vals_tag;
    /* ABI 7, optional, with vals_tag: scratch of at least gnnlm_knn_interp_scratch_bytes(n, k, n_local) bytes (16-byte aligned,
     * contents irrelevant).  The k look-ups of every token are then ROUTED: sorted by region of the tag table (<= 128 KB each) and
     * looked up by one workgroup per region against the region's slice held in LDS, instead of costing one memory request each
     * (the memory system serves ~48 G requests/s whatever their size: 8192 x 1024 one-by-one look-ups take 176 us).  Same result
     * bit for bit (csrc/knn_bucket.hip).  Needs k <= 1024, n <= 2^16, n_local <= 2^27 (else the one-pass kernel runs). */
    void* scratch;  size_t scratch_bytes;
} gnnlm_knn_interp_t;
int gnnlm_knn_interp(const gnnlm_knn_interp_t* desc, void* stream);
size_t gnnlm_knn_interp_scratch_bytes(int64_t n, int32_t k, int64_t n_local);
/* tag[r] = (uint32(vals[r]) * 2654435761) >> 24 for the n rows of a label table (int16 / int32) */
int gnnlm_label_tags(const void* vals, int32_t vals_itemsize, int64_t n, uint8_t* tag, void* stream);

/* ------------------------------------------------------------------------------------------------
 * On-device kNN search, selection half: fold one chunk of scores into a running top-k per query.
 * Replaces the k-selection of faiss `index.search` (knn/knn_model.py:87-101, knn/find_knn.py:55-70); the
 * scores of a chunk come from gnnlm_gemm_nt over a chunk of keys (exact search) or from gnnlm_ivfpq_scan.
 *   value of column c for row r:  v = col_bias[c] + alpha * col_scale[c] * scores[r, c]
 *   (cosine index: col_scale = 1/|key|;  L2: alpha = -2, col_bias = |key|^2, largest = 0, add |q|^2 afterwards)
 * State: best_val / best_id [n, k], best first, ties by ascending id (== a stable argsort of the whole row, so the
 * result is independent of the chunking); unfilled slots hold id -1 like faiss.  No [n, N] matrix is ever built.
 * ---------------------------------------------------------------------------------------------- */
typedef struct gnnlm_topk {
    const float* scores;  int64_t ld;     /* [n, ncols] chunk, row stride ld elements */
    int64_t n;  int32_t ncols;
    int64_t col0;                         /* id of column c = col_ids ? col_ids[c] : col0 + c (col_ids < 0: skipped) */
    const int64_t* col_ids;
    const float* col_scale;  const float* col_bias;  float alpha;   /* alpha 0 is read as 1 */
    int32_t k;                            /* <= 2048 */
    int32_t largest;                      /* 1: keep the k largest values, 0: the k smallest */
    int32_t init;                         /* 1: the state is empty (first chunk), buffers need no initialisation */
    float* best_val;  int64_t* best_id;   /* [n, k] */
    const int32_t* row_ncols;             /* optional [n]: row r only has its first row_ncols[r] columns */
    const int64_t* ids;  int64_t ld_ids;  /* optional per-row ids [n, ncols] (candidate lists of an IVF scan); overrides col0 / col_ids */
} gnnlm_topk_t;
int gnnlm_topk_merge(const gnnlm_topk_t* desc, void* stream);

/* ------------------------------------------------------------------------------------------------
 * IVF-PQ scan with asymmetric distance computation (ADC), inner-product metric with residual codes: the
 * search of the reference's kNN index `OPQ64_1024,IVF4096,PQ64`, nprobe 32 (faiss IndexIVFPQ, by_residual;
 * gnnlm_scripts/wiki103/find_knn.sh:8-13, knn/knn_model.py:100) restated for the GPU:
 *     score(q, x) = <q', c_list(x)> + sum_m lut[q][m][code_m(x)],   lut[q][m][c] = <q'_m, pq_centroid[m][c]>
 * with q' the OPQ-rotated query.  The rotation, the coarse scores and the look-up tables are GEMMs
 * (gnnlm_gemm_nt), the probe selection and the final k-selection are gnnlm_topk_merge; this entry point scans the
 * probed lists.  A task = (query, probe slot); tasks come grouped by list so that concurrent workgroups read the
 * same codes.  Dense mode (tau == NULL): every score of the list is written, for the first probes of a query.
 * Filtered mode: only scores above tau[query] (its current k-th best) are appended to the query's candidate rows.
 * ---------------------------------------------------------------------------------------------- */
typedef struct gnnlm_ivfpq_scan {
    const uint8_t* codes;  const int64_t* ids;  const int64_t* list_off;   /* lists: codes [N, M], ids [N]; list l = [list_off[l], list_off[l+1]) */
    int32_t M;                            /* 8-bit sub-quantizers, M % 16 == 0, M <= 128 */
    const float* lut;  int64_t ld_lut;    /* [n, M*256] */
    const int64_t* probe_list;  const float* probe_bias;  int32_t ld_probe;   /* [n, ld_probe]: probed lists (-1: none) and <q', centroid> */
    const int32_t* task_q;  const int32_t* task_p;  int64_t n_tasks;
    float* out_val;  int64_t* out_id;  int64_t ld_out;  int32_t p0, seg;      /* dense: column (p - p0) * seg + j, ids -1 beyond the list;
                                                                               * out_id NULL (ABI 5): scores only, -inf beyond the list -- the
                                                                               * caller maps the columns it keeps to ids[list_off[list] + j] */
    const float* tau;  float* cand_val;  int64_t* cand_id;  int32_t* cand_cnt;  int32_t cap;   /* filtered: rows of `cap` slots, cand_cnt[q] counts ALL survivors */
    int32_t packed;                       /* ABI 5: nonzero = `codes` is the image of gnnlm_ivfpq_pack_codes and `lut` the tables of
                                           * gnnlm_ivfpq_pack_lut (M = 32 or 64): the bank-conflict-free scan */
    /* L2 metric (faiss METRIC_L2 with residual codes: `IndexBuilder`'s default, knn/index_builder.py:26,118): the score of key x
     * in list l is  -|q' - c_l - r(x)|^2 = probe_bias - sum_m (list_term[l][m][code_m] - 2 lut[q][m][code_m])  with
     * list_term[l][m][c] = |p_mc|^2 + 2 <c_l,m , p_mc>  ([nlist, M * 256], row stride ld_list_term; faiss's "precomputed table")
     * and probe_bias = -|q' - c_l|^2.  list_term != NULL selects it (row-major codes, packed = 0); larger score = nearer */
    const float* list_term;  int64_t ld_list_term;
} gnnlm_ivfpq_scan_t;
int gnnlm_ivfpq_scan(const gnnlm_ivfpq_scan_t* desc, void* stream);

/* The scan's own device layouts (M = 32 or 64).  Codes: blocks of 64 rows stored [M/16 pieces][64 rows][16 B], and inside a
 * row byte s of half h holds sub-quantizer 32 h + (row + s) mod 32 -- `out` has ceil(N / 64) * 64 * M bytes, rows beyond N
 * zero.  Tables: lut [n, M, 256] (row stride ld_lut) -> out [n, M/32, 256, 32].  With these a lane's look-up s goes to
 * sub-quantizer (lane + s) mod 32 and the 32 lanes of an LDS access group always use 32 different banks. */
int gnnlm_ivfpq_pack_codes(const uint8_t* codes, int64_t N, int32_t M, uint8_t* out, void* stream);
int gnnlm_ivfpq_pack_lut(const float* lut, int64_t ld_lut, int64_t n, int32_t M, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ABI 6: the filtered scan on the int8 matrix cores, M = 64 (csrc/ivfpq_mfma.hip) -- the thresholded round of the same
 * search (faiss IndexIVFPQ.search behind knn/knn_model.py:100), as a FILTER with a guaranteed bound followed by an exact
 * float32 re-score, so that the candidates and their scores are those of the one-pass float32 scan above:
 *   gnnlm_ivfpq_pack_tiles     codes [N, 64] -> the scan's image: tiles of 16 rows, [tile][g 0..3][i 0..15][p 0..15] =
 *                              code[16 tile + i][16 g + (i + p) % 16], ceil(N / 16) * 1024 bytes, rows beyond N zero
 *   gnnlm_ivfpq_quantize_lut   lut [n, 64, 256] f32 -> qlut [n][2 halves][256 codes][32 slots] u8 and qmeta [n, 4] f32 =
 *                              {delta, sum_m lo_m, max |lut|, 0} with  lut[m][c] < lo_m + (u + 1) delta  for every entry; the
 *                              table byte is the signed u - 128 (what the i8 matrix instruction reads)
 *   gnnlm_ivfpq_scan8          one workgroup per GROUP = up to 8 queries probing one list; a key (row of the list-ordered
 *                              index) is appended to surv[q] = {row, list} iff the integer sum of its 64 table bytes
 *                              reaches the integer image of tau[q] -- a superset of {score > tau[q]}
 *   gnnlm_ivfpq_tau            the threshold itself: a lower bound of the k-th best score from histograms of the integer sums of
 *                              the first D lists (gnnlm_ivfpq_scan8 with out_hist), replacing the float32 dense round + its k-selection
 *   gnnlm_ivfpq_rescore        exact scores of the survivors (summation order of gnnlm_ivfpq_scan's packed kernel),
 *                              score > tau[q] -> (cand_val, cand_id = payload[row]); cand_cnt[q] counts all of them
 * ---------------------------------------------------------------------------------------------- */
int gnnlm_ivfpq_pack_tiles(const uint8_t* codes, int64_t N, int32_t M, uint8_t* out, void* stream);
int gnnlm_ivfpq_quantize_lut(const float* lut, int64_t ld_lut, int64_t n, int32_t M, uint8_t* qlut, float* qmeta, void* stream);
/* The task table of gnnlm_ivfpq_scan8 from the probe table: (query, probe) pairs of probe_list [n, P] (row stride ld_probe; -1 = none)
 * -> groups of up to 8 queries that probe the same list, in list order: grp_list [G], grp_q [G, 8] (-1 padded), n_groups [1]
 * (device), optionally grp_out [G, 8] = (query * P + probe slot) * seg (the pair's segment of a [n, P, seg] array; NULL: not
 * wanted).  G = n * P / 8 + nlist + 1 entries must be allocated; scratch: 2 * (nlist + 1) int32.  Which queries of a list share a
 * group is not specified (positions come from atomics); the searches' results do not depend on it. */
int gnnlm_ivfpq_build_groups(const int64_t* probe_list, int64_t ld_probe, int64_t n, int32_t P, int32_t nlist, int64_t seg,
                             int32_t* grp_list, int32_t* grp_q, int64_t* grp_out, int32_t* n_groups, int32_t* scratch, void* stream);
typedef struct gnnlm_ivfpq_scan8 {
    const uint8_t* tiles;  const int64_t* list_off;  int32_t M;
    const uint8_t* qlut;  const float* qmeta;
    const float* coarse;  int64_t ld_coarse;          /* [n, nlist] <q', centroid_l>: the bias of list l */
    const float* tau;                                 /* [n] */
    const int32_t* grp_list;  const int32_t* grp_q;   /* [max_groups] list (-1: none), [max_groups, 8] queries (-1: none), sorted by list */
    const int32_t* n_groups;  int32_t max_groups;     /* DEVICE count of groups in use (no host round trip), capacity of the arrays */
    uint32_t* surv;  int32_t* surv_cnt;  int32_t cap; /* [n, cap, 2] {row, list | sum_u << 18}; surv_cnt [n, 16] int32 (one 64-byte line per query: the
                                                       * counters are hammered by atomics), column 0 counts ALL survivors (overflow check) */
    /* threshold pass (out_hist != NULL; tau / surv unused): the integer sums sum_m u (0 .. 255 * 64) of a list's keys are
     * histogrammed per query (1024 bins of 16) and written to out_hist[grp_out[group * 8 + j] .. + 1024) for query j of the group
     * (grp_out < 0: skipped; every (query, list) pair belongs to exactly one group: plain stores) */
    uint32_t* out_hist;  const int64_t* grp_out;
    /* ABI 8, threshold pass only: histogram every sums_stride-th tile of 16 keys of a list (0 / 1: every tile).  A SAMPLE of the
     * keys: the caller asks gnnlm_ivfpq_tau for rank ~k / stride (+ a margin) and must then verify that at least k candidates
     * score above the threshold it got -- gnn-lm_amd/ivfpq.py searches the (rare) queries that fail again with stride 1 */
    int32_t sums_stride;
    /* the workgroups are persistent (one per CU); with work_ctr != NULL ([8, 16] int32, ZEROED by the caller before every call:
     * one counter per XCD, one 64-byte line each) a workgroup fetches its next group from its XCD's counter -- lists of very
     * different lengths stay balanced; NULL: static striding */
    int32_t* work_ctr;
    /* ABI 9: the index's shape, checked -- a survivor record packs the row inside its list into 19 bits and the list into 18, so
     * the filter covers indexes with nlist <= 2^18 lists of fewer than 2^19 keys (max_list = the longest list); longer / more
     * lists are refused here (gnn-lm_amd/ivfpq.py sends such an index to the float32 scan) */
    int32_t nlist;  int64_t max_list;
} gnnlm_ivfpq_scan8_t;
int gnnlm_ivfpq_scan8(const gnnlm_ivfpq_scan8_t* desc, void* stream);
/* tau[q] = a lower bound of query q's k-th best score over its first D probed lists, from the threshold pass's histograms
 * ([n, D, 1024] uint32; list d of query q = probe_list[q, d] with bias probe_bias[q, d]): at least k keys of those lists score
 * above it; -inf if they hold fewer than k keys */
typedef struct gnnlm_ivfpq_tau {
    const uint32_t* hist;  int32_t D;
    const int64_t* probe_list;  const float* probe_bias;  int32_t ld_probe;
    const float* qmeta;
    int64_t n;  int32_t k;
    float* tau;
} gnnlm_ivfpq_tau_t;
int gnnlm_ivfpq_tau(const gnnlm_ivfpq_tau_t* desc, void* stream);
/* ABI 7.  Between the filter and the re-score: a tighter threshold from the survivors themselves.  A survivor record is
 * {row, list | sum_u << 18} with sum_u the key's integer sum (16383: more than the filter's staging could hold): a lower bound of
 * the key's score over ALL probed lists.  tau[q] <- max(tau[q], the k-th largest lower bound of query q's survivors) (at
 * least k keys score above it), the records whose upper bound cannot exceed it are dropped (compacted in place), out_cnt
 * [n, 16] int32 (column 0) = records left; surv_cnt stays what the filter counted.  The search result does not change. */
typedef struct gnnlm_ivfpq_refine {
    uint32_t* surv;  const int32_t* surv_cnt;  int32_t* out_cnt;  int32_t cap;
    float* tau;  const float* qmeta;
    const float* coarse;  int64_t ld_coarse;
    int64_t n;  int32_t k;
} gnnlm_ivfpq_refine_t;
int gnnlm_ivfpq_refine(const gnnlm_ivfpq_refine_t* desc, void* stream);
/* ABI 7.  Results of a search over an index that carries labels (payload = id << label_bits | label, -1 = no result): idx [n]
 * -> the ids in place, out_vals [n] (optional) the labels, `val_last` where there is no result (numpy's vals[-1], knn_model.py:198) */
int gnnlm_ivfpq_split_payload(int64_t* idx, int64_t n, int32_t label_bits, int32_t val_last, int32_t* out_vals, void* stream);
typedef struct gnnlm_ivfpq_rescore {
    const uint8_t* codes;  const int64_t* payload;    /* [N, M] list-ordered codes, [N] what a candidate carries (key id, or id << 24 | label) */
    int32_t M;
    const float* lut;  int64_t ld_lut;                /* [n, M * 256] f32 */
    const float* coarse;  int64_t ld_coarse;
    const float* tau;
    const uint32_t* surv;  const int32_t* surv_cnt;  int32_t cap;
    int64_t n;
    float* cand_val;  int64_t* cand_id;  int32_t* cand_cnt;  int32_t cand_cap;
    /* ABI 10: with `qmeta` (M = 64, k > 0) the refinement of gnnlm_ivfpq_refine runs inside the same launch, ahead of the re-score
     * and under the arrival of the query's table: surv is compacted in place, tau[q] raised in place (both are written through
     * these pointers), out_cnt [n, 16] int32 (column 0; optional) = records left, as there.  NULL: the re-score alone. */
    const float* qmeta;  int32_t k;  int32_t* out_cnt;
} gnnlm_ivfpq_rescore_t;
int gnnlm_ivfpq_rescore(const gnnlm_ivfpq_rescore_t* desc, void* stream);

/* out[0] += sum_i x[i] * (mask ? mask[i] != 0 : 1), accumulated in f64 (score_sum of
 * fairseq_cli/eval_lm.py:273; the reference accumulates in f32 on the CPU) */
int gnnlm_masked_sum_f64(const float* x, const uint8_t* mask, int64_t n, double* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * HGT forward over the implicit token/neighbour graph
 * (TokenGraphTransformerDecoder.extract_graph_features fairseq/models/transformer.py:1011-1053,
 *  HGT.forward fairseq/models/hgt.py:494-513, HGTLayer.forward :299-420).
 * Weights are PREPARED (folded) tensors, see gnnlm_amd/hgt.py::prepare_hgt_weights and DESIGN.md.
 * ---------------------------------------------------------------------------------------------- */
typedef struct gnnlm_hgt_layer {
    /* node type tgt */
    const float *wq_t, *bq_t, *wk_t, *bk_t, *wv_t, *bv_t, *wa_t, *ba_t, *ln_g_t, *ln_b_t;
    /* star edge, absorbed: wku [H, din, dk], wvz_t [H, dk, din], bvz [d] */
    const float *wku, *wvz_t, *bvz;
    int32_t din;
    /* node type ntgt (NULL in the last layer unless ntgt outputs are requested) */
    const float *wq_n, *bq_n, *wk_n, *bk_n, *wv_n, *bv_n, *wa_n, *ba_n, *ln_g_n, *ln_b_n;
    /* ABI 4, layer 0 only, optional (all six or none): the ntgt projections with the OPQ rotation folded in -- [d, M*dsub]
     * weights W' = W . A^T and biases b' = b + W . opq_nba that read the DECODED rows; the rotation (x - b) A itself is then
     * only computed for the rows whose residual the layer needs */
    const float *wq_n0, *bq_n0, *wk_n0, *bk_n0, *wv_n0, *bv_n0;
} gnnlm_hgt_layer_t;

typedef struct gnnlm_hgt {
    int32_t d, n_heads, n_layers, left, right, max_intra_context;
    float ln_eps;
    int32_t M, dsub;
    const float* centroids;    /* [M, 256, dsub] */
    const float* opq_at;       /* [d, M*dsub] = A^T, NULL if the codec has no pre-transform */
    const float* opq_nba;      /* [d] = -(b . A), NULL if no b */
    const uint8_t* codes;  const void* vals;  int32_t vals_itemsize;
    int64_t n_store, row0, n_local;
    const gnnlm_hgt_layer_t* layers;   /* HOST array [n_layers] */
    int32_t gemm_precision;    /* precision of the GEMMs (see gnnlm_gemm_t.precision); 0 = exact f32 */
    const gnnlm_shards_t* shards;  /* ABI 4 (DEVICE pointer): the code table is this set of mapped shards instead of `codes` */
} gnnlm_hgt_t;

typedef struct gnnlm_hgt_io {
    int32_t n_blocks, T, kg;   /* n_blocks independent blocks of T tokens (causal attention stays inside a block) */
    const float* tgt_feats;    /* [n_blocks*T, d] f32 */
    const int64_t* ids;        /* [n_blocks*T, kg] neighbour rows, -1 = none */
    const uint8_t* fetched_codes;   /* optional [n_blocks*T*kg*(1+l+r), M]: slots already fetched (sharded store) */
    const uint8_t* fetched_valid;   /* with fetched_codes: [n_slots] */
    const int32_t* fetched_index;   /* optional: slot s lives in row fetched_index[s] of fetched_codes (the exchange
                                       returns rows bucketed by owner; the permutation is applied by the consumer) */
    int32_t fetched_centres_only;   /* 1: fetched_* hold only the centre slot of each group ([n_blocks*T*kg, M]);
                                       legal only when no ntgt update is needed (n_layers == 1, out_ntgt == NULL) */
    float* out_tgt;            /* [n_blocks*T, d] */
    float* out_ntgt;           /* optional [n_slots, d]: last layer's ntgt states (API parity / tests) */
    uint8_t* out_valid;        /* optional [n_slots] */
    /* ABI 3.  Layer-0 ntgt states given densely instead of as PQ codes: [n_slots, d] rows (row stride ld_ntgt) with
     * their validity bytes [n_slots] -- what HGT.forward has after its input adapters
     * (`F.gelu(adapt_ws[ntype](feat))` when in_dim != hidden_dim, fairseq/models/hgt.py:476-479,505-507), where the
     * decoded features pass a non-linearity and cannot be folded into the star-edge weights.  The code store is then
     * not read; layer 0's `din` must be d. */
    const float* ntgt_feats;  int64_t ld_ntgt;
    const uint8_t* ntgt_valid;
    /* ABI 6, optional: exact de-duplication of context groups (the reference's own "todo: merge same nodes",
     * fairseq/data/token_block_dataset.py:355).  A group's ntgt states depend on its centre row only (ntgt nodes never
     * receive from tgt nodes), so the ntgt pipeline of every layer runs once per DISTINCT centre row of the batch:
     * group_ids [n_unique] the distinct valid rows, group_index [n_blocks*T*kg] the group of neighbour (i, j) (-1: not a
     * neighbour).  Results are those of the un-merged graph.  Not with ntgt_feats / out_ntgt / out_valid.
     * ABI 9: with fetched_codes (sharded store) the fetched_* arrays describe the slots of the GROUPS -- slot g * (1+l+r) + c
     * of group g -- i.e. only the distinct centre rows of the batch were requested from their owners. */
    const int64_t* group_ids;  int64_t n_unique;  const int32_t* group_index;
    /* ABI 7, optional, with group_ids: a CROSS-BATCH cache of the centre states.  The star edges of layer l >= 1 read the
     * centre slot of a group after l ntgt updates, and that state is a function of the centre row alone (fixed weights and
     * store) -- so it is kept between calls: state_cache [n_layers - 1][cache_cap][d] f32, owned by the caller, who also owns
     * the row -> slot map.  Then group_ids / n_unique name only the groups the cache does NOT hold yet (the ntgt pipeline runs
     * over those and their centre states are written to slots group_slot [n_unique]), and group_index [n_blocks*T*kg] holds
     * the cache SLOT of neighbour (i, j) (-1: not a neighbour), new and old alike.  Same kernels, same per-row arithmetic:
     * the result is that of the un-cached call. */
    float* state_cache;  int64_t cache_cap;  const int32_t* group_slot;
    /* ABI 9, optional, with group_ids: the number of groups lives on the DEVICE (int32, written by gnnlm_group_assign on the
     * same stream) -- `n_unique` is then the CAPACITY of group_ids / group_slot (the worst case: every neighbour its own
     * group), which sizes the workspace and the launches; every kernel of the ntgt pipeline reads the count itself, so the
     * caller never synchronises and the call can be captured into a HIP graph. */
    const int32_t* n_unique_dev;
    /* ABI 9, optional, with state_cache and fetched_codes (sharded store): [cache_cap, M] code rows of the cached centres.
     * Layer 0's star edges read the PQ code of EVERY neighbour; with a sharded store only the groups the cache lacks are
     * fetched, so the code row of a centre is kept beside its states (128 B per slot). */
    uint8_t* code_cache;
    /* ABI 11 (optional): layer 0's K / V projections of the ntgt slots keyed by datastore ROW.  A slot's layer-0 K and V are a
     * function of its code row alone, and neighbouring context groups share rows (a group is the window [o - left, o + right]
     * of its centre, token_block_dataset.py:378-400: centres p and p + 1 share 4 of 5 rows -- what consecutive tokens of real
     * kNN-LM retrieval produce).  With `row_table` (DEVICE int32[n_store], all -1 between calls, owned by the caller; needs
     * group_ids and a local or mapped code store) the distinct slot rows of the batch are found on the device
     * (gnnlm_group_assign's claim pass over the slots' rows), decoded and projected ONCE, and the chain attention reads K / V
     * through the slot -> row map.  Same kernels per row: the output is bit-identical. */
    int32_t* row_table;
} gnnlm_hgt_io_t;

/* ------------------------------------------------------------------------------------------------
 * ABI 9.  Group assignment on the device: which DISTINCT centre rows of a batch have to be computed, and where every
 * neighbour finds its group -- what `torch.unique` + a host round trip did before.  No sort, no synchronisation:
 * `slot_of` is a direct row -> slot table (int32 per datastore row, -1 = none, owned by the caller, persistent) claimed
 * with atomicCAS; new groups are appended to group_ids through a device counter in ARRIVAL order (per-row arithmetic
 * does not depend on the position of a row, so results are bit-identical whatever the order).
 *   merge mode (cache_cap == 0): group_ids = the distinct valid rows of `ids`, group_index[e] = position of neighbour e's
 *       row in group_ids (-1: not a neighbour); slot_of is returned clean (all -1).
 *   cache mode (cache_cap > 0): slots are those of a two-generation cache of capacity cache_cap (halves [0, cap/2) and
 *       [cap/2, cap)); group_ids = the rows the cache lacks, group_slot their new slots, group_index[e] = cache slot of
 *       neighbour e.  When the half being filled has no room for the batch's new rows the OTHER half is emptied and
 *       becomes the one being filled (rows this batch found there are computed again); the caller guarantees
 *       n <= cache_cap / 2 so that a fresh half always has room.  cache_state (DEVICE int32[8], zero-initialised by the
 *       caller once): [0] half being filled, [1] / [2] entries of the halves, [3] generation switches, [4] rows computed so far (low 31 bits).
 * counters (DEVICE int32[4], written): [0] = number of groups (pass it as gnnlm_hgt_io_t.n_unique_dev), [1] = 1 if the
 * generation was switched by this call.  group_ids entries beyond the count are -1.
 * Replaces: nothing in the reference (its graph repeats equal nodes, token_block_dataset.py:355 "todo: merge same nodes").
 * ---------------------------------------------------------------------------------------------- */
typedef struct gnnlm_group_assign {
    const int64_t* ids;  int64_t n;      /* neighbour rows of the batch; < 0 or >= n_store: not a neighbour */
    int64_t n_store;
    int32_t* slot_of;                    /* [n_store] */
    int64_t* id_of_slot;                 /* cache mode: [cache_cap] row held by each slot */
    int32_t* cache_state;                /* cache mode: DEVICE int32[8] */
    int64_t cache_cap;
    int64_t* group_ids;                  /* out [n] */
    int32_t* group_slot;                 /* out [n], cache mode */
    int32_t* group_index;                /* out [n] */
    int32_t* counters;                   /* out DEVICE int32[4] */
} gnnlm_group_assign_t;
int gnnlm_group_assign(const gnnlm_group_assign_t* desc, void* stream);

/* x = gelu(x) in place, the exact (erf) form of torch.nn.functional.gelu (input adapters of HGT, hgt.py:507) */
int gnnlm_gelu(float* x, int64_t n, void* stream);
size_t gnnlm_hgt_workspace_bytes(const gnnlm_hgt_t* model, const gnnlm_hgt_io_t* io);
int gnnlm_hgt_forward(const gnnlm_hgt_t* model, const gnnlm_hgt_io_t* io, void* workspace,
                      size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Sharded store, requester side: bucket the requested global rows by owning rank
 * (owner = row / rows_per_rank, out-of-range rows stay on `self_rank`).
 *   counts[world]   rows per owner (device, int64)
 *   send_rows[n]    the rows grouped by owner (order inside a bucket is unspecified)
 *   inv[n]          position of request s inside send_rows (so payload[inv[s]] answers request s)
 * counts must be zeroed by the caller; `cursor` is an int64[world] scratch.
 * ---------------------------------------------------------------------------------------------- */
int gnnlm_bucket_rows(const int64_t* rows, int64_t n, int64_t n_store, int64_t rows_per_rank, int32_t world,
                      int32_t self_rank, int64_t* counts, int64_t* cursor, int64_t* send_rows, int32_t* inv,
                      void* stream);

/* Fixed-capacity variant (no host round trip anywhere in the exchange): owner o's requests are written to
 * send_rows[o*cap, (o+1)*cap), unused slots hold -1 (the call prefills), inv[s] = slot of request s, or world*cap
 * for a request that is not a row of the store or did not fit its bucket (the latter are counted in *overflow,
 * which the call ADDS to).  `cursor` is an int64[world] scratch.  Equal-split all-to-alls can then be used. */
int gnnlm_bucket_rows_padded(const int64_t* rows, int64_t n, int64_t n_store, int64_t rows_per_rank, int32_t world,
                             int64_t cap, int64_t* cursor, int64_t* send_rows, int32_t* inv, int64_t* overflow, void* stream);

/* Peer-mapped alternative to the exchange: every rank maps the shards of its peers into its own address
 * space (hipIpcOpenMemHandle; over xGMI the loads of this kernel then go to the owner's HBM directly) and gathers the
 * requested rows itself -- one kernel, no collective, no bucketing, payload in request order.  shard[g] is the base of the
 * rows [shard_row0[g], shard_row0[g] + shard_rows[g]) rank g HOLDS (its range plus halo, if any); the owner of a row is
 * row / rows_per_rank as in gnnlm_bucket_rows.  Rows outside [0, n_store) (the -1 of invalid slots) give zero rows and
 * out_valid 0.  row_bytes: 1..16 or a multiple of 16 (128-B code rows, 4-B / 2-B labels). */
typedef struct gnnlm_peer_gather {
    const void* shard[16];
    int64_t shard_row0[16], shard_rows[16];
    int32_t world, row_bytes;
    int64_t rows_per_rank, n_store;
    const int64_t* rows;  int64_t n;
    void* out;                 /* [n, row_bytes] */
    uint8_t* out_valid;        /* optional [n] */
} gnnlm_peer_gather_t;
int gnnlm_gather_rows_peer(const gnnlm_peer_gather_t* desc, void* stream);
/* hipDeviceEnablePeerAccess(peer) for the current device (no error if already enabled); the Python side calls it once per
 * peer before the first gather when the shards live on other devices */
int gnnlm_enable_peer_access(int32_t peer_device);

/* ------------------------------------------------------------------------------------------------
 * Opt-in live timing (bench.py's roofline): while a profile is open every launch of the selected
 * kernels is bracketed by HIP events on its own stream.  One profile at a time; launches from several host threads
 * may be recorded into it (records and pools are mutex-guarded).
 * ---------------------------------------------------------------------------------------------- */
typedef struct gnnlm_profile_entry {
    int32_t kernel_id;
    int64_t launches;
    double total_ms;           /* sum of event-measured launch durations */
    double flops, bytes;       /* algorithmic work of those launches */
} gnnlm_profile_entry_t;
const char* gnnlm_kernel_name(int32_t kernel_id);
int gnnlm_profile_begin(uint32_t kernel_mask);      /* bit i selects kernel id i */
int gnnlm_profile_end(gnnlm_profile_entry_t* out, int32_t n_max, int32_t* n_out);   /* synchronises the events */

/* ------------------------------------------------------------------------------------------------
 * Owning HBM store for non-torch callers (DataStore / PlasmaArray residency,
 * knn/data_store.py:29-64, fairseq/tasks/language_modeling.py:274-276).  These DO allocate and the
 * uploads synchronise the given stream before returning.
 * ---------------------------------------------------------------------------------------------- */
typedef struct gnnlm_store gnnlm_store_t;
int gnnlm_store_create(int64_t n_store, int64_t row0, int64_t n_local, int32_t M, int32_t vals_itemsize,
                       int32_t device, gnnlm_store_t** out);
int gnnlm_store_upload_codes(gnnlm_store_t* s, const uint8_t* host_codes, int64_t first_local_row, int64_t n_rows, void* stream);
int gnnlm_store_upload_vals(gnnlm_store_t* s, const void* host_vals, int64_t first_local_row, int64_t n_rows, void* stream);
const uint8_t* gnnlm_store_codes(const gnnlm_store_t* s);
const void* gnnlm_store_vals(const gnnlm_store_t* s);
int gnnlm_store_destroy(gnnlm_store_t* s);

#ifdef __cplusplus
}
#endif
#endif /* GNNLM_H */
