"""torch-tensor wrappers over the kernel-level C ABI (include/gnnlm.h).

torch is plumbing here (device memory + the current HIP stream); every function below launches a
hand-written gfx950 kernel through libgnnlm_hip.so and raises if that library is missing.
"""
import ctypes

import torch

from . import _lib
from ._lib import call, call_desc, ptr, stream


def _dev(*ts):
    """Refuse host tensors explicitly: the product path has no CPU fallback."""
    for t in ts:
        if t is not None and not (t.is_cuda and t.is_contiguous() or (t.is_cuda and t.dim() == 2 and t.stride(1) == 1)):
            raise _lib.GnnlmError("gnnlm_amd kernels need contiguous device (HIP) tensors; there is no CPU fallback")


def _f32(*ts):
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise TypeError(f"expected float32, got {t.dtype}")
    return ts[0]


def _dtype(t, dt, what):
    """The C ABI reinterprets raw pointers: a wrong dtype must fail here, not be silently re-read."""
    if t is not None and t.dtype != dt:
        raise TypeError(f"{what}: expected {dt}, got {t.dtype}")


PRECISIONS = {"f32": 0, "bf16x3": 1, "bf16x6": 2}


def gemm_nt(A, W, bias=None, residual=None, alpha=1.0, out=None, bias_mode=1, gate=None,
            a_rows=None, m_dev=None, precision=0, c_rows=None):
    """C = alpha * A @ W.T (+ gate*bias) (+ residual).  A [M,K] (row stride may exceed K), W [N,K].
    a_rows: logical row r reads A[a_rows[r]] (< 0: zero row); c_rows: row r is stored to (and its residual
    read from) row c_rows[r] of ``out`` (which must then be given)."""
    _f32(A, W, bias, residual, gate, out)
    _dtype(a_rows, torch.int32, "a_rows"), _dtype(m_dev, torch.int32, "m_dev")
    _dev(A, W, bias, residual, gate, a_rows, m_dev, out, c_rows)
    M, K = A.shape
    N = W.shape[0]
    if a_rows is not None:
        M = a_rows.shape[0]
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    g = _lib.gnnlm_gemm_t()
    g.A, g.lda = A.data_ptr(), A.stride(0)
    g.W, g.ldw = W.data_ptr(), W.stride(0)
    g.C, g.ldc = out.data_ptr(), out.stride(0)
    if a_rows is not None:
        g.a_rows = a_rows.data_ptr()
        g.a_rows_bound = A.shape[0]                    # every gathered row lies inside A
    if c_rows is not None:
        assert c_rows.dtype == torch.int32 and c_rows.shape[0] == M
        g.c_rows = c_rows.data_ptr()
    if bias is not None:
        g.bias, g.bias_mode = bias.data_ptr(), bias_mode
    if gate is not None:
        g.gate = gate.data_ptr()
    if residual is not None:
        g.R, g.ldr = residual.data_ptr(), residual.stride(0)
    if m_dev is not None:
        g.m_dev = m_dev.data_ptr()
    g.alpha = alpha
    g.M, g.N, g.K = M, N, K
    g.precision = PRECISIONS.get(precision, precision)
    call_desc("gnnlm_gemm_nt", g)
    return out


def gemm_lse(A, W, pick=None, alpha=1.0, m_dev=None, precision=0):
    """lse[r] = logsumexp_n(alpha * A[r] . W[n]) and picked[r] = alpha * A[r] . W[pick[r]], without
    materialising the [M, N] logits (LSE epilogue of the GEMM + gnnlm_lse_reduce)."""
    _f32(A, W)
    _dtype(m_dev, torch.int32, "m_dev")
    _dev(A, W, pick, m_dev)
    M, K = A.shape
    N = W.shape[0]
    n_parts = 2 * ((N + 127) // 128)
    part = torch.empty(M, n_parts, 2, device=A.device, dtype=torch.float32)
    lse = torch.empty(M, device=A.device, dtype=torch.float32)
    picked = torch.zeros(M, device=A.device, dtype=torch.float32)
    g = _lib.gnnlm_gemm_t()
    g.A, g.lda, g.W, g.ldw = A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0)
    g.lse_part = part.data_ptr()
    if pick is not None:
        assert pick.dtype == torch.int32
        g.lse_pick, g.lse_picked = pick.data_ptr(), picked.data_ptr()
    if m_dev is not None:
        g.m_dev = m_dev.data_ptr()
    g.alpha = alpha
    g.M, g.N, g.K = M, N, K
    g.precision = PRECISIONS.get(precision, precision)
    call_desc("gnnlm_gemm_nt", g)
    call("gnnlm_lse_reduce", ptr(part), n_parts, M, ptr(m_dev), ptr(lse), stream())
    return lse, picked


def pq_gather_decode(codes, centroids, ids, left=0, right=0, n_store=None, row0=0, vals=None,
                     want_x=True, want_codes=False, want_labels=False, want_valid=True):
    """Gather + decode the slots of each centre row in ``ids`` (flattened).  Returns a dict."""
    _dev(codes, centroids, ids, vals)
    M, ksub, dsub = centroids.shape
    assert ksub == 256 and codes.dtype == torch.uint8 and ids.dtype == torch.int64
    n_g = 1 + left + right
    G = ids.numel()
    S = G * n_g
    dev = codes.device
    d = _lib.gnnlm_gather_t()
    d.codes = codes.data_ptr()
    d.n_local = codes.shape[0]
    d.n_store = n_store if n_store is not None else codes.shape[0]
    d.row0 = row0
    d.M, d.dsub = M, dsub
    d.centroids = centroids.data_ptr()
    d.ids, d.n_groups = ids.data_ptr(), G
    d.left, d.right = left, right
    d.vals_itemsize = 4
    out = {}
    if vals is not None:
        d.vals, d.vals_itemsize = vals.data_ptr(), vals.element_size()
    if want_x:
        out["x"] = torch.empty(S, M * dsub, device=dev, dtype=torch.float32)
        d.out_x, d.ld_x = out["x"].data_ptr(), M * dsub
    if want_codes:
        out["codes"] = torch.empty(S, M, device=dev, dtype=torch.uint8)
        d.out_codes = out["codes"].data_ptr()
    if want_labels:
        out["labels"] = torch.empty(S, device=dev, dtype=torch.int32)
        d.out_labels = out["labels"].data_ptr()
    if want_valid:
        out["valid"] = torch.empty(S, device=dev, dtype=torch.uint8)
        d.out_valid = out["valid"].data_ptr()
    call_desc("gnnlm_pq_gather_decode", d)
    return out


def pq_lookup_direct(codes, centroids):
    """x[n, m*dsub:(m+1)*dsub] = centroids[m, codes[n, m]] for an explicit code matrix (no store)."""
    _dev(codes, centroids)
    n, M = codes.shape
    dsub = centroids.shape[2]
    x = torch.empty(n, M * dsub, device=codes.device, dtype=torch.float32)
    valid = torch.ones(n, device=codes.device, dtype=torch.uint8)
    d = _lib.gnnlm_gather_t()
    d.codes, d.direct, d.in_valid = codes.data_ptr(), 1, valid.data_ptr()
    d.M, d.dsub, d.centroids = M, dsub, centroids.data_ptr()
    d.n_groups, d.vals_itemsize = n, 4
    d.out_x, d.ld_x = x.data_ptr(), M * dsub
    call_desc("gnnlm_pq_gather_decode", d)
    return x


def star_attn(U, ids, codes=None, centroids=None, row0=0, X=None, x_group_stride=1, codes_direct=0, n_store=None,
              nb_valid=None, nb_valid_stride=1, shards=None, rows_per_rank=0):
    """n_store: rows of the whole store (ids >= n_store are not neighbours; default: the rows of ``codes``);
    nb_valid (uint8): validity byte of neighbour (i, j) at [(i*kg + j) * nb_valid_stride];
    shards = [(codes_g [rows_g, M], first global row), ...] + rows_per_rank: a range-sharded table whose shards are all
    mapped into this process (instead of ``codes``; needs ``n_store``)."""
    _dev(U, ids, codes, centroids, X, nb_valid)
    _f32(U, centroids, X)
    _dtype(ids, torch.int64, "ids"), _dtype(codes, torch.uint8, "codes"), _dtype(nb_valid, torch.uint8, "nb_valid")
    T, H, D = U.shape
    kg = ids.shape[1]
    Z = torch.empty_like(U)
    has_nb = torch.empty(T, device=U.device, dtype=torch.float32)
    a = _lib.gnnlm_star_attn_t()
    a.U, a.ids = U.data_ptr(), ids.data_ptr()
    a.T, a.H, a.D, a.kg = T, H, D, kg
    if shards:
        from .hgt import CodeStore, shards_device_ptr
        keep = CodeStore(codes=None, centroids=centroids, n_store=n_store, shards=shards, rows_per_rank=rows_per_rank)
        a.shards = shards_device_ptr(keep)
        a.M, a.dsub = centroids.shape[0], centroids.shape[2]
        a.centroids = centroids.data_ptr()
    elif codes is not None:
        a.codes, a.row0, a.n_local = codes.data_ptr(), row0, codes.shape[0]
        a.M, a.dsub = centroids.shape[0], centroids.shape[2]
        a.centroids = centroids.data_ptr()
        a.codes_direct = codes_direct
        if n_store is None and not codes_direct:
            n_store = row0 + codes.shape[0]
    else:
        a.X, a.ldx, a.x_group_stride = X.data_ptr(), X.stride(0), x_group_stride
    a.Z, a.has_nb = Z.data_ptr(), has_nb.data_ptr()
    a.n_store = n_store or 0
    if nb_valid is not None:
        a.nb_valid, a.nb_valid_stride = nb_valid.data_ptr(), nb_valid_stride
    call_desc("gnnlm_star_attn", a)
    if shards:
        Z._gnnlm_keep = keep                                    # the device copy of the shard table outlives the queued launch
    return Z, has_nb


def chain_attn(Q, K, V, valid, left, right, H, scale=None):
    _dev(Q, K, V, valid, scale)
    S, d = Q.shape
    n_g = 1 + left + right
    out = torch.empty_like(Q)
    c = _lib.gnnlm_chain_attn_t()
    c.Q, c.K, c.V, c.ld = Q.data_ptr(), K.data_ptr(), V.data_ptr(), Q.stride(0)
    c.valid = valid.data_ptr()
    c.n_groups, c.left, c.right, c.H, c.dk = S // n_g, left, right, H, d // H
    if scale is not None:
        c.scale = scale.data_ptr()
    c.out, c.ldo = out.data_ptr(), out.stride(0)
    call_desc("gnnlm_chain_attn", c)
    return out


def causal_attn(Q, K, V, n_blocks, T, H, max_ctx=0):
    """Fused causal attention (T = 256, d_k = 128): Q, K', V' [n_blocks*T, H*dk] -> [n_blocks*T, H*dk]."""
    _f32(Q), _f32(K), _f32(V)
    _dev(Q, K, V)
    d = Q.shape[1]
    out = torch.empty_like(Q)
    call("gnnlm_causal_attn", ptr(Q), ptr(K), ptr(V), Q.stride(0), ptr(out), out.stride(0), n_blocks, T, H, d // H,
         max_ctx, stream())
    return out


def causal_softmax_(S, T, max_ctx=0):
    """In place over S [n_mats, T, ld]."""
    n_mats, T_, ld = S.shape
    assert T_ == T
    call("gnnlm_causal_softmax", ptr(S), n_mats, T, ld, max_ctx, stream())
    return S


def layernorm(x, gamma, beta, eps=1e-5, valid=None):
    rows, d = x.shape
    out = torch.empty_like(x)
    call("gnnlm_layernorm", ptr(x), x.stride(0), ptr(gamma), ptr(beta), ptr(out), out.stride(0), rows, d, eps,
         ptr(valid), stream())
    return out


def gelu_(x):
    """In place, exact (erf) GELU == torch.nn.functional.gelu's default."""
    _dev(x)
    _f32(x)
    assert x.is_contiguous()
    call("gnnlm_gelu", ptr(x), x.numel(), stream())
    return x


def half_to_float(x):
    out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    call("gnnlm_half_to_float", ptr(x), ptr(out), x.numel(), stream())
    return out


def filter_neighbors(ids, tok_pos, invalid_ctx, out=None):
    """``--invalid-neighbor-context`` (token_block_dataset.py:360-362): ids [n, kg] i64, tok_pos [n] i64 -> ids with the
    neighbours inside their own token's context replaced by -1 (the graph consumers skip -1 ids)."""
    _dev(ids, tok_pos)
    _dtype(ids, torch.int64, "ids"), _dtype(tok_pos, torch.int64, "tok_pos")
    n, kg = ids.shape
    assert tok_pos.shape == (n,) and ids.is_contiguous() and tok_pos.is_contiguous()
    out = torch.empty_like(ids) if out is None else out
    call("gnnlm_filter_neighbors", ptr(ids), ptr(tok_pos), n, kg, int(invalid_ctx), ptr(out), stream())
    return out


def row_lse_pick(logits, pick=None):
    rows, n = logits.shape
    lse = torch.empty(rows, device=logits.device, dtype=torch.float32)
    picked = torch.empty(rows, device=logits.device, dtype=torch.float32)
    call("gnnlm_row_lse_pick", ptr(logits), logits.stride(0), rows, None, n, ptr(pick), ptr(lse), ptr(picked), stream())
    return lse, picked


_TAG_TABLES = {}                     # id(vals) -> (weakref to vals, version, tag table)
TAG_TABLE_MIN_ROWS = 1 << 20         # below this the label table sits in the L2 anyway


def label_tags(vals, build=True):
    """The one-byte tag table of a label table (``gnnlm_label_tags``; see ``gnnlm_knn_interp_t.vals_tag``), built once per
    table and kept while the table lives (keyed on the tensor object and its in-place version counter).  ``build=False``:
    only what is cached (inside a stream capture nothing may be allocated or launched on the side)."""
    import weakref
    for k_ in [k_ for k_, (ref, _, _) in _TAG_TABLES.items() if ref() is None]:
        del _TAG_TABLES[k_]
    hit = _TAG_TABLES.get(id(vals))
    if hit is not None and hit[0]() is vals and hit[1] == vals._version:
        return hit[2]
    if not build:
        return None
    assert vals.dim() == 1 and vals.is_contiguous() and vals.dtype in (torch.int16, torch.int32)
    tag = torch.empty(vals.shape[0], dtype=torch.uint8, device=vals.device)
    call("gnnlm_label_tags", ptr(vals), vals.element_size(), vals.shape[0], ptr(tag), stream())
    _TAG_TABLES[id(vals)] = (weakref.ref(vals), vals._version, tag)
    return tag


_KNN_SCRATCH = {}                    # (device, stream) -> scratch of the routed look-ups (knn_bucket.hip)
_KNN_SCRATCH_CAPTURED = []           # scratch buffers whose addresses live in captured HIP graphs: kept for good
KNN_BUCKET_MIN_LOOKUPS = 1 << 20     # below this the one-pass kernel is as fast (three launches against one)


def knn_interp(lm_logp, sims, ids, targets, temperature, lmbda, vals=None, n_store=None, row0=0, knn_vals=None, vals_tag="auto",
               bucketed="auto"):
    """``vals_tag``: "auto" = gather through the one-byte tag table (built on first use) when the label table is large and
    k fits the register kernel; True / False force it on / off (tests, A/B).  ``bucketed``: "auto" = with the tag table and at
    least 2^20 look-ups, route the look-ups region by region through the L2 (gnnlm_knn_interp_t.scratch); True / False force."""
    _dev(lm_logp, sims, ids, targets, vals, knn_vals)
    _f32(lm_logp, sims)
    _dtype(ids, torch.int64, "ids"), _dtype(targets, torch.int64, "targets")
    if vals is not None and vals.dtype not in (torch.int16, torch.int32):
        raise TypeError(f"vals: expected int16 / int32, got {vals.dtype}")
    n, k = sims.shape
    dev = sims.device
    out = torch.empty(n, device=dev, dtype=torch.float32)
    pk = torch.empty(n, device=dev, dtype=torch.float32)
    rec = torch.empty(n, device=dev, dtype=torch.int64)
    d = _lib.gnnlm_knn_interp_t()
    d.lm_logp, d.sims, d.ids, d.targets = lm_logp.data_ptr(), sims.data_ptr(), ids.data_ptr(), targets.data_ptr()
    d.vals_itemsize = 4
    if vals is not None:
        d.vals, d.vals_itemsize = vals.data_ptr(), vals.element_size()
        d.n_local = vals.shape[0]
        d.n_store = n_store if n_store is not None else vals.shape[0]
        d.row0 = row0
    if knn_vals is not None:
        assert knn_vals.dtype == torch.int32
        d.knn_vals = knn_vals.data_ptr()
    elif vals is not None and vals.dim() == 1 and vals.is_contiguous() and k <= 1024 and \
            (vals_tag is True or (vals_tag == "auto" and vals.shape[0] >= TAG_TABLE_MIN_ROWS)):
        capturing = torch.cuda.is_current_stream_capturing()
        tag = label_tags(vals, build=not capturing)
        if tag is not None:
            d.vals_tag = tag.data_ptr()
            if bucketed is True or (bucketed == "auto" and n * k >= KNN_BUCKET_MIN_LOOKUPS):
                need = _lib.lib().gnnlm_knn_interp_scratch_bytes(n, k, vals.shape[0])
                key = (str(dev), _lib.raw_stream(dev))
                sc = _KNN_SCRATCH.get(key)
                if (sc is None or sc.numel() < need) and not capturing and n <= (1 << 16) and vals.shape[0] <= (1 << 27):
                    sc = _KNN_SCRATCH[key] = torch.empty(need, dtype=torch.uint8, device=dev)
                if sc is not None and sc.numel() >= need:
                    d.scratch, d.scratch_bytes = sc.data_ptr(), sc.numel()
                    if capturing and not any(t is sc for t in _KNN_SCRATCH_CAPTURED):
                        # its address is baked into a HIP graph: a later, larger request replaces the dictionary entry, but this
                        # buffer must outlive every replay (never freed)
                        _KNN_SCRATCH_CAPTURED.append(sc)
    d.n, d.k = n, k
    d.temperature, d.lmbda = temperature, lmbda
    d.out_logp, d.out_pknn, d.out_recall = out.data_ptr(), pk.data_ptr(), rec.data_ptr()
    call_desc("gnnlm_knn_interp", d)
    return out, pk, rec


def topk_merge(scores, best_val, best_id, col0=0, col_ids=None, col_scale=None, col_bias=None, alpha=1.0, largest=True,
               init=False, row_ncols=None, ids=None):
    """Fold the score chunk ``scores`` [n, ncols] into the running top-k state (``best_val`` f32 / ``best_id`` i64
    [n, k], best first): value of column c = col_bias[c] + alpha * col_scale[c] * scores[:, c], id = col_ids[c] or
    col0 + c.  In place; ``init=True`` treats the state as empty (first chunk)."""
    _dev(scores, best_val, best_id, col_ids, col_scale, col_bias, row_ncols)
    _f32(scores, best_val, col_scale, col_bias)
    _dtype(best_id, torch.int64, "best_id"), _dtype(col_ids, torch.int64, "col_ids"), _dtype(row_ncols, torch.int32, "row_ncols")
    n, ncols = scores.shape
    assert best_val.shape == best_id.shape and best_val.shape[0] == n and best_val.is_contiguous() and best_id.is_contiguous()
    d = _lib.gnnlm_topk_t()
    d.scores, d.ld, d.n, d.ncols = scores.data_ptr(), scores.stride(0), n, ncols
    d.col0 = col0
    if col_ids is not None:
        d.col_ids = col_ids.data_ptr()
    if col_scale is not None:
        d.col_scale = col_scale.data_ptr()
    if col_bias is not None:
        d.col_bias = col_bias.data_ptr()
    if row_ncols is not None:
        d.row_ncols = row_ncols.data_ptr()
    if ids is not None:                                  # per-row ids [n, ncols]
        _dev(ids)
        _dtype(ids, torch.int64, "ids")
        assert ids.shape == scores.shape
        d.ids, d.ld_ids = ids.data_ptr(), ids.stride(0)
    d.alpha, d.k, d.largest, d.init = alpha, best_val.shape[1], int(largest), int(init)
    d.best_val, d.best_id = best_val.data_ptr(), best_id.data_ptr()
    call_desc("gnnlm_topk_merge", d)
    return best_val, best_id


def ivfpq_pack_tiles(codes):
    """codes [N, 64] u8 -> the int8-MFMA scan's image (csrc/ivfpq_mfma.hip): ceil(N / 16) tiles x [4 groups][16 rows][16 bytes],
    byte p of (tile t, group g, row i) = codes[16 t + i][16 g + (i + p) % 16]; rows beyond N are zero."""
    _dev(codes)
    _dtype(codes, torch.uint8, "codes")
    N, M = codes.shape
    out = torch.empty(-(-N // 16) * 16 * M, dtype=torch.uint8, device=codes.device)
    call("gnnlm_ivfpq_pack_tiles", ptr(codes), N, M, ptr(out), stream())
    return out


def ivfpq_quantize_lut(lut, M=64):
    """lut [n, M * 256] f32 -> (qlut [n, 2, 256, 32] u8, qmeta [n, 4] f32 = {delta, sum_m lo_m, max |lut|, 0}) with
    lut[m][c] < lo_m + (u + 1) * delta for every entry, u = qlut[m // 32][c][m % 32] ^ 0x80 (the table stores the signed
    byte u - 128, what the i8 matrix instruction reads) -- the filter's one-sided bound."""
    _dev(lut)
    _f32(lut)
    n = lut.shape[0]
    qlut = torch.empty(n, 2, 256, 32, dtype=torch.uint8, device=lut.device)
    qmeta = torch.empty(n, 4, dtype=torch.float32, device=lut.device)
    call("gnnlm_ivfpq_quantize_lut", ptr(lut), lut.stride(0), n, M, ptr(qlut), ptr(qmeta), stream())
    return qlut, qmeta


def masked_sum_f64(x, mask=None, acc=None):
    if acc is None:
        acc = torch.zeros(1, device=x.device, dtype=torch.float64)
    call("gnnlm_masked_sum_f64", ptr(x), ptr(mask), x.numel(), ptr(acc), stream())
    return acc
