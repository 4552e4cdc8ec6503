"""IVF-PQ index searched on the GPU -- the on-device replacement of the reference's faiss CPU search
(``knn/knn_model.py:87-101``: ``index.search`` on ``faiss_store.cosine`` = ``OPQ64_1024,IVF4096,PQ64``, nprobe 32,
``gnnlm_scripts/wiki103/find_knn.sh:8-13``; built by ``knn/index_builder.py``).

The algorithm is faiss's published IVFADC with residual codes and an OPQ pre-rotation, inner-product metric:

    x' = R x                       (cosine index: x is L2-normalised first, index_builder.py:90-95,118)
    list(x) = argmax_l <x', c_l>   residual r = x' - c_list,  code_m = argmin_c |r_m - p_mc|^2
    score(q, x) = <q', c_list(x)> + sum_m <q'_m, p_{m, code_m(x)}>         (ADC, look-up table per query)
    search: the nprobe lists with the largest <q', c_l>, then the k best scores among their members.

Search path (everything on the device, nothing [n, N]-sized): rotation, coarse scores and look-up tables on the f32
MFMA GEMM (``gnnlm_gemm_nt``), probe selection and k-selection by ``gnnlm_topk_merge``, the list scan by
``gnnlm_ivfpq_scan`` in two rounds -- the best ``dense_probes`` lists of every query are scored in full and give its
k-th-best threshold, the remaining lists only emit scores above that threshold.

At M = 64 (the reference's PQ64) the thresholded round is a FILTER on the int8 matrix cores followed by an exact re-score
(``csrc/ivfpq_mfma.hip``: 8-bit tables with a guaranteed one-sided bound, eight queries of a list per workgroup, the
sums taken by ``v_smfmac_i32_16x16x128_i8`` (round 3: ``v_mfma_i32_16x16x64_i8``); survivors re-scored in float32 in the summation order of the float32 scan), so
candidates and scores are those of the one-pass float32 scan -- ``GNNLM_IVF_SCAN=f32`` selects that scan for A/B runs.
With ``attach_vals`` the index carries each key's label next to its id (one 8-byte payload per key), and the search
returns ``vals[ids]`` with the neighbours: the label gather of knn/knn_model.py:198 disappears.

``IVFPQIndex.build`` is the offline producer (the reference delegates it to faiss: index_builder.py:79-150): plain
Lloyd k-means for the product quantizers, spherical k-means (unit-norm centroids, what faiss's index_factory sets for
inner-product indexes) for the coarse one, a random orthonormal matrix for R (faiss alternates OPQ updates; any
orthonormal R gives a valid index).  It runs on the GPU with torch ops -- offline tooling, not the hot path.

PARITY UNPINNED against faiss itself (not in the image, no version pinned): the pin is the numpy restatement
``oracle/ivfpq.py`` over the SAME index arrays (tests/test_knn_search_gpu.py)."""
import contextlib
import ctypes
import os

import numpy as np
import torch

from . import _lib, ops


class _LazyStats(dict):
    """Work counters of a search whose device-side terms are only reduced when somebody reads them: a search enqueues no
    bookkeeping kernels for numbers that only bench.py and the tests look at (``add`` keeps the tensors and a thunk)."""

    def add(self, key, thunk):
        dict.setdefault(self, key, 0)
        self.__dict__.setdefault("_pending", {}).setdefault(key, []).append(thunk)

    def _resolve(self, key):
        pend = self.__dict__.get("_pending", {}).pop(key, None)
        if pend:
            dict.__setitem__(self, key, dict.__getitem__(self, key) + sum(t() for t in pend))

    def __getitem__(self, key):
        self._resolve(key)
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def items(self):
        return [(k, self[k]) for k in list(self.keys())]

    def clone(self):
        out = _LazyStats()
        for k, v in self.items():
            dict.__setitem__(out, k, v.clone() if torch.is_tensor(v) else v)
        return out


class _PendingSearch:
    """Handle of ``IVFPQIndex.search_begin``."""

    def __init__(self, index, q, k, query_block, return_vals, val, idx, over, worst, ev, cap):
        self.index, self.q, self.k, self.query_block, self.return_vals = index, q, k, query_block, return_vals
        self.val, self.idx, self.over, self.worst, self.ev = val, idx, over, worst, ev
        self.cap = cap                                   # survivor capacity the buffers of THIS search were allocated with
        self._out = None

    def result(self):
        if self._out is None:
            self._out = self.index._finish(self)
        return self._out


def _kmeans(x, k, iters, gen, spherical=False, init=None):
    """Lloyd's algorithm (squared L2) on the device; empty clusters are re-seeded from random points.  ``spherical``: the
    centroids are L2-normalised after every update (faiss ClusteringParameters.spherical, set by index_factory for
    METRIC_INNER_PRODUCT coarse quantizers).  ``init``: centroids to start from (OPQ's inner PQ rounds)."""
    n = x.shape[0]
    if init is not None:
        cen = init.clone()
    else:
        cen = x[torch.randperm(n, generator=gen, device=x.device)[:k]].clone()
    if cen.shape[0] < k:
        cen = torch.cat([cen, cen[torch.randint(0, cen.shape[0], (k - cen.shape[0],), generator=gen, device=x.device)]])
    for _ in range(iters):
        assign = torch.empty(n, dtype=torch.int64, device=x.device)
        c2 = (cen ** 2).sum(1)
        for s in range(0, n, 1 << 18):
            xs = x[s:s + (1 << 18)]
            assign[s:s + (1 << 18)] = (c2[None, :] - 2 * xs @ cen.t()).argmin(1)
        cnt = torch.bincount(assign, minlength=k)
        new = torch.zeros_like(cen).index_add_(0, assign, x)
        live = cnt > 0
        new[live] /= cnt[live, None].to(x.dtype)
        dead = (~live).nonzero().reshape(-1)
        if dead.numel():
            new[dead] = x[torch.randint(0, n, (dead.numel(),), generator=gen, device=x.device)]
        cen = new / new.norm(dim=1, keepdim=True).clamp_min(1e-20) if spherical else new
    return cen


def _pq_assign(x, cen):
    """x [n, dsub], cen [256, dsub] -> nearest centroid per row (squared L2), in chunks."""
    out = torch.empty(x.shape[0], dtype=torch.int64, device=x.device)
    c2 = (cen ** 2).sum(1)
    for s in range(0, x.shape[0], 1 << 18):
        out[s:s + (1 << 18)] = (c2[None, :] - 2 * x[s:s + (1 << 18)] @ cen.t()).argmin(1)
    return out


def train_opq(x, M, R0, iters, gen, pq_iters=4):
    """The OPQ rotation (Ge et al., "Optimized Product Quantization", the non-parametric solution; what faiss's OPQMatrix trains
    for the ``OPQ64_1024`` block of the reference's index type, knn/index_builder.py:60-64): alternate (a) a product quantizer of
    M x 256 centroids on the rotated vectors x R^T (k-means per sub-space, warm-started from the previous round) and (b) the
    orthogonal Procrustes solution R^T = U V^T, U S V^T = x^T y, y = the quantizer's reconstruction of x R^T -- the rotation
    under which the quantizer loses least.  x [n, d] f32 on the device, R0 [d_out, d] with orthonormal rows; -> R [d_out, d]."""
    n, d = x.shape
    dsub = R0.shape[0] // M                                                   # R0 [d_out, d]: d_out < d for an `OPQ<M>_<d_out>` block that reduces
    R, cents = R0.clone(), [None] * M
    for _ in range(iters):
        xr = x @ R.t()
        y = torch.empty_like(xr)
        for m in range(M):
            sub = xr[:, m * dsub:(m + 1) * dsub].contiguous()
            cents[m] = _kmeans(sub, 256, pq_iters, gen, init=cents[m])
            y[:, m * dsub:(m + 1) * dsub] = cents[m][_pq_assign(sub, cents[m])]
        U, _, Vt = torch.linalg.svd((x.t() @ y).double(), full_matrices=False)
        R = (U @ Vt).t().to(torch.float32).contiguous()
    return R


def build_groups(pl, nlist, seg=None):
    """(query, probe) pairs of ``pl`` [nq, P] (list ids, -1 = none) -> groups of up to 8 queries that probe the same list,
    sorted by list: grp_list [G] int32, grp_q [G, 8] int32 (-1 padded), the number of groups in use as a one-element int32
    tensor on the device.  Torch ops on the tensor's device, no host round trip; G is the upper bound pairs / 8 + nlist + 1.
    With ``seg``: also grp_out [G, 8] int64, the offset (query * P + probe slot) * seg of the pair's segment in a [nq, P, seg]
    array (-1: none).  What gnnlm_ivfpq_scan8 takes as its task table (include/gnnlm.h)."""
    nq, P = pl.shape
    dev = pl.device
    n = nq * P
    key = torch.where(pl < 0, torch.full_like(pl, nlist), pl).reshape(-1)
    skey, order = torch.sort(key, stable=True)
    cnt = torch.zeros(nlist + 1, dtype=torch.int64, device=dev).scatter_add_(0, key, torch.ones_like(key))   # pairs per list (last bucket: no list); no host sync, unlike bincount
    start = torch.cumsum(cnt, 0) - cnt
    gcnt = (cnt + 7) // 8
    goff = torch.cumsum(gcnt, 0) - gcnt                                       # first group of every list
    pos = torch.arange(n, device=dev) - start[skey]                           # position inside the run of one list
    gid = goff[skey] + pos // 8
    G = n // 8 + nlist + 1
    none = skey >= nlist                                                      # pairs without a list: their groups sort last and are
    grp_list = torch.full((G,), -1, dtype=torch.int32, device=dev)            # cut off by n_groups, their entries stay -1
    grp_list[gid] = torch.where(none, torch.full_like(skey, -1), skey).to(torch.int32)
    grp_q = torch.full((G, 8), -1, dtype=torch.int32, device=dev)
    grp_q[gid, pos % 8] = torch.where(none, torch.full_like(order, -1), torch.div(order, P, rounding_mode="floor")).to(torch.int32)
    n_groups = gcnt[:nlist].sum().reshape(1).to(torch.int32)                  # (the groups of the "no list" bucket sort last: cut off)
    if seg is None:
        return grp_list, grp_q, n_groups, G
    grp_out = torch.full((G, 8), -1, dtype=torch.int64, device=dev)
    grp_out[gid, pos % 8] = torch.where(none, torch.full_like(order, -1), order * seg)   # order = query * P + slot
    return grp_list, grp_q, n_groups, G, grp_out


class IVFPQIndex:
    UNDERFLOW = 1 << 30      # "survivor count" of a query whose sampled threshold left fewer than k candidates: searched again like an overflow
    COLUMNS = (1 << 30) - 1  # ... of a query with more candidates than the selection has columns (a larger survivor capacity would not help it)
    """faiss ``search`` contract: ``search(queries [n, d] f32, k) -> (scores [n, k] descending, ids [n, k], -1 padded)``."""

    LABEL_BITS = 24                                                          # payload = id << 24 | label (ids < 2^39, labels < 2^24)

    def __init__(self, R, coarse, pq, list_off, list_ids, list_codes, nprobe=32, cosine=True, dense_probes=None, cand_cap=None,
                 score_bytes=6 << 30, scan=None, metric="ip", list_term_bytes=4 << 30):
        self.R, self.coarse, self.pq = R, coarse, pq                         # [d, d], [nlist, d], [M, 256, dsub]  f32
        self.list_off, self.list_ids, self.list_codes = list_off, list_ids, list_codes   # i64 [nlist+1], i64 [N], u8 [N, M]
        self.nprobe, self.cosine, self.dense_probes, self.cand_cap = nprobe, cosine, dense_probes, cand_cap
        # every S-th tile of the threshold lists is histogrammed (GNNLM_IVF_SAMPLE; 1: all of them -- see _search_block_mfma)
        self.threshold_sample = max(1, int(os.environ.get("GNNLM_IVF_SAMPLE", "4")))
        self.sample_min_keys_per_k = 64          # ... when the threshold lists hold at least this many keys per neighbour asked for
        self.sample_sigmas = 4.5                 # margin of the sample's rank (tests lower it to force the verification to fail)
        self.sample_fail_frac = 0.01             # more failing queries than this in a batch: the index stops sampling
        self.score_bytes = score_bytes                                       # budget of the dense round's score rows per query block
        self.payload, self.has_vals, self.val_last = list_ids, False, 0      # what a candidate carries through the selection
        self.device = R.device
        if metric not in ("ip", "l2") or (metric == "l2" and cosine):
            raise ValueError("IVFPQIndex: metric is 'ip' (cosine: inner product of normalised vectors) or 'l2'")
        self.metric = metric
        self.d, self.nlist = coarse.shape[1], coarse.shape[0]
        self.M, _, self.dsub = pq.shape
        self.ntotal = list_ids.shape[0]
        self.max_list = int((list_off[1:] - list_off[:-1]).max().item()) if self.nlist else 0
        # M = 32 / 64: the scan runs on its own image of the code rows (blocks of 64 rows, bytes in rotated order) and on
        # tables in [half][code][sub-quantizer] order -- look-ups without LDS bank conflicts (csrc/ivfpq.hip)
        # M = 64: the int8-MFMA search's image of the code rows (tiles of 16 rows, rotated byte order; csrc/ivfpq_mfma.hip)
        self.packed_codes = self.tiles = None
        scan = scan or os.environ.get("GNNLM_IVF_SCAN", "mfma")              # "f32": the float32 scan everywhere (A/B, tests)
        self.list_term = None
        if metric == "l2":
            # squared distances with residual codes: |q' - c_l - r|^2 = |q' - c_l|^2 + sum_m (T[l][m][code] - 2 <q'_m, p_mc>) with the
            # per-list table T[l][m][c] = |p_mc|^2 + 2 <c_l,m, p_mc> (faiss IndexIVFPQ's precomputed table); the scan adds it to the
            # query's table while it fills LDS (gnnlm_ivfpq_scan, list_term).  nlist * M KiB of HBM: built when that is affordable
            need = self.nlist * self.M * 1024
            if need > list_term_bytes:
                raise ValueError(f"IVFPQIndex: the L2 metric keeps a [nlist, M, 256] float32 table ({need / 2**30:.1f} GiB here, "
                                 f"list_term_bytes = {list_term_bytes / 2**30:.1f} GiB)")
            self.coarse_n2 = (coarse ** 2).sum(1).contiguous()
            cross = torch.einsum("lmd,mcd->lmc", coarse.reshape(self.nlist, self.M, self.dsub), pq)
            self.list_term = ((pq ** 2).sum(2)[None] + 2.0 * cross).reshape(self.nlist, self.M * 256).contiguous()
            scan = "rowmajor"
        if scan == "rowmajor":
            pass                                                             # the row-major kernels (one table per (query, list) task)
        elif self.M == 64 and self.ntotal and self.ntotal < (1 << 32) and scan != "f32" and self.max_list < (1 << 19) and self.nlist <= (1 << 18):
            # (a survivor record packs the row inside its list into 19 bits and the list into 18: longer / more lists take the float32 scan)
            self.tiles = ops.ivfpq_pack_tiles(self.list_codes)
        elif self.M in (32, 64) and self.ntotal:
            self.packed_codes = torch.empty(-(-self.ntotal // 64) * 64 * self.M, dtype=torch.uint8, device=self.device)
            _lib.call("gnnlm_ivfpq_pack_codes", _lib.ptr(self.list_codes), self.ntotal, self.M, _lib.ptr(self.packed_codes), _lib.stream())
        # lists per query behind the threshold: the float32 dense round scores 2 in full; the int8 threshold pass is cheap and its
        # bound a little loose, 6 lists give the tighter threshold (fewer survivors to re-score) for less time
        if self.dense_probes is None:
            self.dense_probes = 6 if self.tiles is not None else 2
        # records per query: the int8 filter's SURVIVORS (8 bytes each; a sampled threshold lets ~8 k of them through on average and twice that
        # where list lengths vary: 32768 keeps the re-searches of overflowing queries rare), the float32 scan's candidates
        if self.cand_cap is None:
            self.cand_cap = 32768 if self.tiles is not None else 16384
        self.refine_tau = os.environ.get("GNNLM_IVF_REFINE", "1") != "0"     # gnnlm_ivfpq_refine between the filter and the re-score (A/B: 0)
        self.fuse_refine = os.environ.get("GNNLM_IVF_FUSED", "1") != "0"     # ... inside the re-score's launch (A/B: 0 = two launches)
        self.fork_tables = os.environ.get("GNNLM_IVF_FORK", "1") != "0"      # ADC tables on a side stream beside the coarse scores (A/B: 0)
        self._side_streams = {}                                              # raw stream -> its side stream
        self.stats = {}                                                      # device-side work counters of the last search (bench.py)

    def attach_vals(self, vals):
        """Carry the keys' labels with the index: payload[r] = id << 24 | vals[id] for the key at list position r (the reference
        reads ``self.vals[knns]`` after the search, knn/knn_model.py:198: k random 4-byte reads per query).  ``vals``: the
        datastore's label table on the device ([N] or [N, 1], int16 / int32); -1 ids of the result read ``vals[-1]`` as numpy's
        wrap-around does there."""
        v = vals.reshape(-1).to(self.device)
        if self.ntotal == 0:
            return self
        if int(v.max().item()) >= (1 << self.LABEL_BITS) or int(v.min().item()) < 0 or int(self.list_ids.max().item()) >= (1 << (63 - self.LABEL_BITS)):
            raise ValueError("attach_vals: labels must fit 24 bits and key ids 39 bits")
        self.payload = (self.list_ids << self.LABEL_BITS) | v[self.list_ids].to(torch.int64)
        self.has_vals, self.val_last = True, int(v[-1].item())
        return self

    # ------------------------------------------------------------------------------------------ offline producer
    @classmethod
    def build(cls, keys, nlist, M, device="cuda", cosine=True, nprobe=32, iters=10, train_size=262144, seed=0, chunk=1 << 18,
              opq_iters=0, metric="ip", **kw):
        """Train (rotation, coarse centroids, residual product quantizer) on a sample and add every key.  ``opq_iters`` = 0: a
        random orthonormal rotation (spreads the variance over the sub-spaces); > 0: that many rounds of OPQ training from it
        (``train_opq``: what the reference's ``OPQ64_1024`` block asks faiss for)."""
        device = torch.device(device)
        gen = torch.Generator(device=device)
        gen.manual_seed(seed)
        N, d = keys.shape
        assert d % M == 0 and (d // M) % 4 == 0 and M % 16 == 0, "need dsub % 4 == 0 and M % 16 == 0"
        dsub = d // M

        def rows(lo, hi):
            x = torch.from_numpy(np.ascontiguousarray(keys[lo:hi]).astype(np.float32)).to(device) if not isinstance(keys, torch.Tensor) \
                else keys[lo:hi].to(device, torch.float32)
            return x / (x ** 2).sum(1, keepdim=True).sqrt() if cosine else x

        cpu_gen = torch.Generator().manual_seed(seed)
        R = torch.linalg.qr(torch.randn(d, d, generator=cpu_gen, dtype=torch.float64))[0].to(torch.float32).to(device)
        pick = np.sort(np.random.RandomState(seed).choice(N, size=min(N, train_size), replace=False))
        xt = torch.cat([rows(int(s), int(min(N, s + chunk)))[torch.from_numpy(pick[(pick >= s) & (pick < s + chunk)] - s).to(device)]
                        for s in range(0, N, chunk)])
        if opq_iters > 0:
            R = train_opq(xt, M, R, opq_iters, gen)
        xt = xt @ R.t()
        l2 = metric == "l2"
        assert not (l2 and cosine), "the L2 index takes the keys as they are"
        coarse = _kmeans(xt, nlist, iters, gen, spherical=cosine)
        # the list of a vector: the centroid with the largest inner product (IndexFlatIP quantizer) / the nearest one (IndexFlatL2)
        nearest = lambda x: ((x @ coarse.t()) - (0.5 * (coarse ** 2).sum(1)[None, :] if l2 else 0.0)).argmax(1)
        resid = xt - coarse[nearest(xt)]
        pq = torch.stack([_kmeans(resid[:, m * dsub:(m + 1) * dsub].contiguous(), 256, iters, gen) for m in range(M)])
        # add every key: list assignment + residual PQ codes (HIP argmin kernel of TorchPQCodec.encode)
        assign = torch.empty(N, dtype=torch.int64, device=device)
        codes = torch.empty(N, M, dtype=torch.uint8, device=device)
        norm2 = (pq ** 2).sum(2).contiguous()
        for s in range(0, N, chunk):
            x = rows(s, min(N, s + chunk)) @ R.t()
            a = nearest(x)
            assign[s:s + chunk] = a
            r = (x - coarse[a]).contiguous()
            _lib.call("gnnlm_pq_encode", _lib.ptr(r), r.stride(0), _lib.ptr(pq), _lib.ptr(norm2), M, dsub, r.shape[0],
                      _lib.ptr(codes[s:s + chunk]), _lib.stream())
        order = torch.sort(assign, stable=True).indices
        off = torch.zeros(nlist + 1, dtype=torch.int64, device=device)
        off[1:] = torch.cumsum(torch.bincount(assign, minlength=nlist), 0)
        return cls(R.contiguous(), coarse.contiguous(), pq.contiguous(), off, order.contiguous(), codes[order].contiguous(),
                   nprobe=nprobe, cosine=cosine, metric=metric, **kw)

    def save(self, path):
        np.savez(path, **{k: getattr(self, k).cpu().numpy() for k in ("R", "coarse", "pq", "list_off", "list_ids", "list_codes")},
                 meta=np.array([self.nprobe, int(self.cosine), int(self.metric == "l2")]))

    @classmethod
    def load(cls, path, device="cuda", **kw):
        z = np.load(path)
        t = lambda k: torch.from_numpy(z[k]).to(device)
        kw.setdefault("nprobe", int(z["meta"][0]))
        kw.setdefault("metric", "l2" if len(z["meta"]) > 2 and z["meta"][2] else "ip")
        return cls(t("R"), t("coarse"), t("pq"), t("list_off"), t("list_ids"), t("list_codes"), cosine=bool(z["meta"][1]), **kw)

    @classmethod
    def from_faiss_file(cls, path, device="cuda", cosine=True, **kw):
        """The reference's own index file (``faiss.write_index`` of ``OPQ64_1024,IVF4096,PQ64``, knn/index_builder.py:79-150;
        ``--index-file``) read without faiss (faiss_io.read_ivfpq_index) -- inner-product indexes with residual codes."""
        from . import faiss_io
        z = faiss_io.read_ivfpq_index(path)
        if z["metric"] != z["coarse_metric"] or not z["by_residual"]:
            raise ValueError(f"{path}: the on-device search covers IVF-PQ with residual codes whose coarse quantizer has the index's "
                             f"metric (what knn/index_builder.py builds), found metric={z['metric']} coarse={z['coarse_metric']} "
                             f"by_residual={z['by_residual']}")
        kw.setdefault("metric", z["metric"])
        if z["metric"] == "l2":
            cosine = False
        d = z["coarse"].shape[1]
        R = z["R"] if z["R"] is not None else np.eye(d, dtype=np.float32)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        kw.setdefault("nprobe", z["nprobe"])
        return cls(t(R), t(z["coarse"]), t(z["pq"]), t(z["list_off"]), t(z["list_ids"]), t(z["list_codes"]), cosine=cosine, **kw)

    # ------------------------------------------------------------------------------------------ search
    def _scan(self, lut, probe_val, probe_id, p_lo, p_hi, out=None, tau=None, cand=None, cap=None):
        n = lut.shape[0]
        dev = self.device
        pl = probe_id[:, p_lo:p_hi]
        # tasks grouped by list (bookkeeping on the device, no host round trip): concurrent workgroups share code rows
        order = torch.sort(pl.reshape(-1), stable=True).indices
        task_q = torch.div(order, p_hi - p_lo, rounding_mode="floor").to(torch.int32)
        task_p = (order % (p_hi - p_lo) + p_lo).to(torch.int32)
        s = _lib.gnnlm_ivfpq_scan_t()
        s.codes, s.ids, s.list_off, s.M = self.list_codes.data_ptr(), self.payload.data_ptr(), self.list_off.data_ptr(), self.M
        if self.packed_codes is not None:
            s.codes, s.packed = self.packed_codes.data_ptr(), 1                 # `lut` is then the packed table set
        s.lut, s.ld_lut = lut.data_ptr(), lut.stride(0)
        s.probe_list, s.probe_bias, s.ld_probe = probe_id.data_ptr(), probe_val.data_ptr(), probe_id.stride(0)
        s.task_q, s.task_p, s.n_tasks = task_q.data_ptr(), task_p.data_ptr(), order.numel()
        if self.list_term is not None:
            s.list_term, s.ld_list_term = self.list_term.data_ptr(), self.list_term.stride(0)
        if tau is None:
            s.out_val, s.ld_out, s.p0, s.seg = out.data_ptr(), out.stride(0), p_lo, self.max_list       # scores only (out_id NULL)
        else:
            s.tau, s.cand_val, s.cand_id, s.cand_cnt, s.cap = tau.data_ptr(), cand[0].data_ptr(), cand[1].data_ptr(), cand[2].data_ptr(), cap
        _lib.call_desc("gnnlm_ivfpq_scan", s)

    def _groups(self, pl, seg=None):
        """The scan's task table on the device (gnnlm_ivfpq_build_groups: histogram, prefix, scatter -- four small launches instead
        of the ~25 of ``build_groups``, the torch reference of the same table); same tuple."""
        if not pl.is_cuda or os.environ.get("GNNLM_IVF_TORCH_GROUPS"):
            return build_groups(pl, self.nlist, seg)
        nq, P = pl.shape
        dev = pl.device
        G = nq * P // 8 + self.nlist + 1
        grp_list = torch.empty(G, dtype=torch.int32, device=dev)
        grp_q = torch.empty(G, 8, dtype=torch.int32, device=dev)
        grp_out = torch.empty(G, 8, dtype=torch.int64, device=dev) if seg is not None else None
        n_groups = torch.empty(1, dtype=torch.int32, device=dev)
        scratch = torch.empty(2 * (self.nlist + 1), dtype=torch.int32, device=dev)
        _lib.call("gnnlm_ivfpq_build_groups", ctypes.c_void_p(pl.data_ptr()), pl.stride(0), nq, P, self.nlist, seg or 0, _lib.ptr(grp_list),
                  _lib.ptr(grp_q), _lib.ptr(grp_out), _lib.ptr(n_groups), _lib.ptr(scratch), _lib.stream())
        return (grp_list, grp_q, n_groups, G) if seg is None else (grp_list, grp_q, n_groups, G, grp_out)

    def search_device(self, q, k, query_block=None, return_vals=False):
        """The search, on device tensors: (scores [n, k] descending, ids [n, k], -1 padded[, vals [n, k] int32 with
        ``attach_vals``]).  The thresholded round keeps at most ``cand_cap`` survivors per query; the survivor counts are read
        back once per call (the only host sync).  Queries with more (a query whose dense lists hold fewer than k keys has no
        threshold) are searched again on their own with every probed list scored in full; if many overflow, the capacity is
        doubled for good and the call repeated."""
        return self.search_begin(q, k, query_block, return_vals).result()

    def search_begin(self, q, k, query_block=None, return_vals=False):
        """Enqueue the search and return a handle; ``handle.result()`` waits for the SEARCH only (an event behind its last kernel, the
        worst survivor count on its way to pinned memory) and returns what ``search_device`` returns.  Work enqueued between the two
        calls (the language model's softmax, which does not depend on the neighbours) keeps the device busy while the host looks at
        the count and enqueues what follows: no pipeline bubble behind the search's one host round trip."""
        q = q.to(self.device, torch.float32).contiguous()
        self.stats = _LazyStats(pairs=0, survivors=0, candidates=0, queries=q.shape[0], M=self.M, requeried=0)
        cap = self.cand_cap                                                  # the capacity THIS search's buffers were allocated with
        val, idx, over = self._search_once(q, k, query_block, self.dense_probes, cap)
        worst = ev = None
        if over is not None:
            if getattr(self, "_worst_host", None) is None:                    # pinned landing slots, one per search in flight (a ring of 8)
                self._worst_host, self._worst_next = torch.empty(8, dtype=torch.int32).pin_memory(), 0
            worst = self._worst_host[self._worst_next:self._worst_next + 1]
            self._worst_next = (self._worst_next + 1) % 8
            worst.copy_(over.max().reshape(1), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        return _PendingSearch(self, q, k, query_block, return_vals, val, idx, over, worst, ev, cap)

    def _finish(self, h):
        q, k, query_block = h.q, h.k, h.query_block
        val, idx, over = h.val, h.idx, h.over
        cap = h.cap                                                           # (not self.cand_cap: another search in flight may have raised it since)
        while over is not None:
            if h.ev is not None:
                h.ev.synchronize()
                if int(h.worst[0]) <= cap:
                    break
                h.ev = None
            bad = (over > cap).nonzero().reshape(-1)                          # host sync
            if bad.numel() == 0:
                break
            n_under = int((over[bad] >= self.UNDERFLOW).sum().item())
            n_cols = int((over[bad] == self.COLUMNS).sum().item())            # too many CANDIDATES for the selection's columns: searched again one by one
            if n_under > self.sample_fail_frac * q.shape[0] and self.threshold_sample > 1:   # the sample does not stand for its lists on this data: stop sampling
                self.threshold_sample = 1
                self.stats = _LazyStats(pairs=0, survivors=0, candidates=0, queries=q.shape[0], M=self.M, requeried=0)
                val, idx, over = self._search_once(q, k, query_block, self.dense_probes, cap)
                continue
            if (bad.numel() - n_under - n_cols) * 8 > q.shape[0] and cap < (1 << 18):
                cap *= 2
                self.cand_cap = max(self.cand_cap, cap)
                self.stats = _LazyStats(pairs=0, survivors=0, candidates=0, queries=q.shape[0], M=self.M, requeried=0)
                val, idx, over = self._search_once(q, k, query_block, self.dense_probes, cap)
                continue
            sub, cap2 = q[bad].contiguous(), cap
            main = self.stats.clone()
            while True:                                                       # every probed list in the threshold / dense round
                v2, i2, o2 = self._search_once(sub, k, query_block, self.nprobe, cap2)
                if o2 is None or int(o2.max().item()) <= cap2:
                    break
                if cap2 >= (1 << 22):                                         # survivors beyond the capacity would be dropped silently
                    raise _lib.GnnlmError(f"ivfpq search: {int(o2.max().item())} survivors of one query exceed the largest candidate "
                                          f"capacity ({cap2}); lower k / nprobe or search this index with scan='f32'")
                cap2 *= 2
            val[bad], idx[bad] = v2, i2
            dict.__setitem__(main, "requeried", int(bad.numel()))             # (the counters describe the main pass)
            self.stats = main
            break
        self._overflow = None
        if self.metric == "l2":
            val = -val                                                        # scores are -distance: squared distances, ascending, +inf padded
        if not self.has_vals:
            return (val, idx, None) if h.return_vals else (val, idx)
        # payload -> (id in place, label): one pass (gnnlm_ivfpq_split_payload) instead of five elementwise torch kernels over [n, k]
        vals = torch.empty(idx.shape, dtype=torch.int32, device=idx.device) if h.return_vals else None
        _lib.call("gnnlm_ivfpq_split_payload", _lib.ptr(idx), idx.numel(), self.LABEL_BITS, self.val_last, _lib.ptr(vals), _lib.stream())
        return (val, idx, vals) if h.return_vals else (val, idx)

    def _search_once(self, q, k, query_block, dense_probes, cap):
        n, dev = q.shape[0], self.device
        nprobe = min(self.nprobe, self.nlist)
        dense = max(1, min(dense_probes, nprobe))
        val = torch.empty(n, k, device=dev, dtype=torch.float32)
        idx = torch.empty(n, k, device=dev, dtype=torch.int64)
        if query_block is None:                                               # groups of 8 queries per list want many queries per block
            # (the int8 filter groups 8 queries per list: the more queries a block holds, the fuller its groups and the longer a list's bytes
            # stay in the XCD's L2 -- per 8192 queries 12.2 ms in blocks of 8192, 11.4 of 16384, 11.0 of 32768; ~18 GB of temporaries at 32768)
            query_block = int(os.environ.get("GNNLM_IVF_QUERY_BLOCK", "32768")) if self.tiles is not None else 1024
        # the dense round's score rows: query_block * dense * max_list floats, bounded (a skewed index has long lists)
        qb = query_block if self.tiles is not None else max(1, min(query_block, self.score_bytes // max(1, 4 * dense * max(self.max_list, 1))))
        over = None
        for q0 in range(0, n, qb):
            o = self._search_block(q[q0:q0 + qb], k, val[q0:q0 + qb], idx[q0:q0 + qb], nprobe, dense, cap)
            if o is not None:
                over = o if over is None else torch.cat([over, o])
        return val, idx, over

    def _search_block(self, qs, k, bv, bi, nprobe, dense, cap):
        dev = self.device
        nq = qs.shape[0]
        qr = ops.gemm_nt(qs, self.R)                                            # q' = R q
        # the ADC tables (a store-bound batched GEMM, 2 KB per query and sub-quantizer) and their 8-bit images depend on q' alone: they
        # are built on a side stream beside the coarse scores / probe selection / task table of this stream (GNNLM_IVF_FORK=0: in line)
        lut, tables, side = self._tables_begin(qr)
        cs = ops.gemm_nt(qr, self.coarse)                                       # <q', c_l>
        pv = torch.empty(nq, nprobe, device=dev, dtype=torch.float32)
        pi = torch.empty(nq, nprobe, device=dev, dtype=torch.int64)
        if self.metric == "l2":                                                 # the nprobe NEAREST lists: argmax <q', c> - |c|^2 / 2
            ops.topk_merge(cs, pv, pi, largest=True, init=True, col_bias=-0.5 * self.coarse_n2)
            pv = (2.0 * pv - (qr ** 2).sum(1, keepdim=True)).contiguous()      # the list's bias: -|q' - c_l|^2
        else:
            ops.topk_merge(cs, pv, pi, largest=True, init=True)                 # the nprobe best lists, best first
        self.stats.add("pairs", lambda pi=pi: (self.list_off[1:] - self.list_off[:-1])[pi.clamp(min=0)].masked_fill(pi < 0, 0).sum().double())
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)                       # (lut / tables were allocated on THIS stream: no record_stream needed)
        # (the int8 path: `cap` bounds a query's SURVIVORS of the filter; what scores above the refined threshold afterwards is a fraction of them
        # and gets 16384 columns at most -- the width the one-chunk k-selection takes)
        ccap = min(cap, 16384) if self.tiles is not None else cap
        cv = torch.empty(nq, ccap, device=dev, dtype=torch.float32)
        ci = torch.empty(nq, ccap, device=dev, dtype=torch.int64)
        cc = torch.zeros(nq, device=dev, dtype=torch.int32)
        if self.tiles is not None:
            return self._search_block_mfma(k, bv, bi, nprobe, dense, cs, pv, pi, lut, cv, ci, cc, cap, tables)
        lut_s = lut
        if self.packed_codes is not None:
            lut_s = torch.empty_like(lut)
            _lib.call("gnnlm_ivfpq_pack_lut", _lib.ptr(lut), lut.stride(0), nq, self.M, _lib.ptr(lut_s), _lib.stream())
        # round 1: the best `dense` lists of every query, every score
        ov = torch.empty(nq, dense * self.max_list, device=dev, dtype=torch.float32)
        self._scan(lut_s, pv, pi, 0, dense, out=ov)
        ops.topk_merge(ov, bv, bi, largest=True, init=True)                     # ids = columns of ov
        # the columns kept -> payloads: column = probe slot * max_list + position in the list (-inf: beyond a list)
        lst = torch.gather(pi, 1, torch.div(bi, self.max_list, rounding_mode="floor").clamp_(0, dense - 1))
        pos = self.list_off[lst.clamp(min=0)] + bi % self.max_list
        bi.copy_(torch.where(torch.isinf(bv) | (lst < 0), torch.full_like(bi, -1), self.payload[pos.clamp_(0, max(self.ntotal - 1, 0))]))
        if nprobe <= dense:
            return None
        # round 2: the other lists only emit scores above the query's k-th best so far
        tau = torch.where(bi[:, k - 1] >= 0, bv[:, k - 1], torch.full_like(bv[:, k - 1], float("-inf"))).contiguous()
        self._scan(lut_s, pv, pi, dense, nprobe, tau=tau, cand=(cv, ci, cc), cap=cap)
        self.stats.add("candidates", lambda cc=cc: cc.sum().double())
        if getattr(self, "keep_candidates", False):                          # tests / debugging: the round-2 candidates of the last block
            self.last_candidates = (cv, ci, cc, tau)
        ops.topk_merge(cv, bv, bi, ids=ci, largest=True, init=False, row_ncols=cc.clamp(max=cap))
        return cc

    def _scan8(self, qlut, qmeta, cs, groups, tau=None, surv=None, hist=None, stride=1):
        d = _lib.gnnlm_ivfpq_scan8_t()
        d.sums_stride = stride
        d.tiles, d.list_off, d.M = self.tiles.data_ptr(), self.list_off.data_ptr(), self.M
        d.nlist, d.max_list = self.nlist, self.max_list
        d.qlut, d.qmeta, d.coarse, d.ld_coarse = qlut.data_ptr(), qmeta.data_ptr(), cs.data_ptr(), cs.stride(0)
        d.grp_list, d.grp_q, d.n_groups, d.max_groups = groups[0].data_ptr(), groups[1].data_ptr(), groups[2].data_ptr(), groups[3]
        ctr = torch.zeros(8, 16, device=self.device, dtype=torch.int32)          # the persistent workgroups' work counters (one per XCD)
        d.work_ctr = ctr.data_ptr()
        if getattr(self, "keep_work_ctr", False):                              # instrumented builds (GNNLM_IVF8_EXP & 512) leave phase times here
            self.last_work_ctr = ctr
        if hist is not None:
            d.out_hist, d.grp_out = hist.data_ptr(), groups[4].data_ptr()
        else:
            d.tau, d.surv, d.surv_cnt, d.cap = tau.data_ptr(), surv[0].data_ptr(), surv[1].data_ptr(), surv[0].shape[1]
        _lib.call_desc("gnnlm_ivfpq_scan8", d)

    def _tables_begin(self, qr):
        """lut[q][m][c] = <q'_m, p_mc> (M small GEMMs) and, for the int8 filter, its byte image: enqueued on a side stream of the
        current one when the filter will run (the caller joins before the first consumer).  -> (lut, (qlut, qmeta) | None, side | None)"""
        nq, dev = qr.shape[0], self.device
        lut = torch.empty(nq, self.M * 256, device=dev, dtype=torch.float32)
        tables = None
        if self.tiles is not None:
            tables = (torch.empty(nq, 2, 256, 32, dtype=torch.uint8, device=dev), torch.empty(nq, 4, dtype=torch.float32, device=dev))
        side = None
        if self.tiles is not None and self.fork_tables and not torch.cuda.is_current_stream_capturing():
            cur = torch.cuda.current_stream()
            side = self._side_streams.get(cur.cuda_stream)
            if side is None:
                side = self._side_streams[cur.cuda_stream] = torch.cuda.Stream(device=dev)
            side.wait_stream(cur)
        with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
            g = _lib.gnnlm_gemm_t()
            g.A, g.lda, g.W, g.ldw, g.C, g.ldc = qr.data_ptr(), qr.stride(0), self.pq.data_ptr(), self.dsub, lut.data_ptr(), self.M * 256
            g.M, g.N, g.K, g.batch1 = nq, 256, self.dsub, self.M
            g.sA1, g.sW1, g.sC1 = self.dsub, 256 * self.dsub, 256
            _lib.call_desc("gnnlm_gemm_nt", g)
            if tables is not None:
                _lib.call("gnnlm_ivfpq_quantize_lut", _lib.ptr(lut), lut.stride(0), nq, self.M, _lib.ptr(tables[0]), _lib.ptr(tables[1]), _lib.stream())
        return lut, tables, side

    def _search_block_mfma(self, k, bv, bi, nprobe, dense, cs, pv, pi, lut, cv, ci, cc, cap, tables=None):
        """M = 64: everything on the int8 matrix cores (csrc/ivfpq_mfma.hip).  (1) threshold pass: histograms of the integer sums of the
        first `dense` lists -> a lower bound tau of the query's k-th best score (no per-key output, no selection); (2) filter: every probed
        list, keys whose integer sum can reach tau; (3) exact float32 scores of the survivors, score > tau -> candidates;
        (4) one k-selection over the candidates."""
        dev = self.device
        nq, ccap = pv.shape[0], cv.shape[1]
        qlut, qmeta = tables if tables is not None else ops.ivfpq_quantize_lut(lut, self.M)
        hist = torch.empty(nq, dense, 1024, device=dev, dtype=torch.int32)      # per (query, list): sum_u >> 4 counted on the device
        # The threshold pass histograms a SAMPLE of its lists' keys (every S-th tile of 16) when they hold plenty of them: tau is then the
        # bound of rank k / S + 4.5 sigma of the sample (sigma = sqrt(k (S - 1)) / S: the k best land in the sample binomially), i.e. with
        # probability ~1e-5 per query fewer than k keys score above it.  Nothing is taken on trust: a query with fewer than k candidates
        # above its threshold is reported like an overflowing one and searched again with every probed list in an exact threshold pass.
        S = self.threshold_sample if (dense < nprobe and self.ntotal / max(self.nlist, 1) * dense >= self.sample_min_keys_per_k * k) else 1
        rank = k if S == 1 else max(1, min(k, -(-k // S) + int(np.ceil(self.sample_sigmas * np.sqrt(k * (S - 1)) / S))))
        self._scan8(qlut, qmeta, cs, self._groups(pi[:, :dense], seg=1024), hist=hist, stride=S)
        tau = torch.empty(nq, device=dev, dtype=torch.float32)
        t = _lib.gnnlm_ivfpq_tau_t()
        t.hist, t.D = hist.data_ptr(), dense
        t.probe_list, t.probe_bias, t.ld_probe = pi.data_ptr(), pv.data_ptr(), pi.stride(0)
        t.qmeta, t.n, t.k, t.tau = qmeta.data_ptr(), nq, rank, tau.data_ptr()
        _lib.call_desc("gnnlm_ivfpq_tau", t)
        surv = torch.empty(nq, cap, 2, device=dev, dtype=torch.int32)
        sc16 = torch.zeros(nq, 16, device=dev, dtype=torch.int32)              # one 64-byte line per counter (column 0)
        g2 = self._groups(pi)
        self.stats.add("groups", lambda g=g2[2]: g[0].double())
        self._scan8(qlut, qmeta, cs, g2, tau=tau, surv=(surv, sc16))
        sc = sc16[:, 0]
        # a tighter threshold from the survivors' own integer sums (all lists, un-binned), and only the survivors that can beat it
        rc16 = sc16
        fused = self.refine_tau and self.fuse_refine                          # refinement inside the re-score's launch (ABI 10)
        if fused:
            rc16 = torch.empty_like(sc16)
            self.stats.add("rescored", lambda rc=rc16: rc[:, 0].sum().double())
        elif self.refine_tau:
            rc16 = torch.empty_like(sc16)
            f = _lib.gnnlm_ivfpq_refine_t()
            f.surv, f.surv_cnt, f.out_cnt, f.cap = surv.data_ptr(), sc16.data_ptr(), rc16.data_ptr(), cap
            f.tau, f.qmeta, f.coarse, f.ld_coarse, f.n, f.k = tau.data_ptr(), qmeta.data_ptr(), cs.data_ptr(), cs.stride(0), nq, k
            _lib.call_desc("gnnlm_ivfpq_refine", f)
            self.stats.add("rescored", lambda rc=rc16: rc[:, 0].sum().double())
        r = _lib.gnnlm_ivfpq_rescore_t()
        r.codes, r.payload, r.M = self.list_codes.data_ptr(), self.payload.data_ptr(), self.M
        r.lut, r.ld_lut, r.coarse, r.ld_coarse, r.tau = lut.data_ptr(), lut.stride(0), cs.data_ptr(), cs.stride(0), tau.data_ptr()
        r.surv, r.surv_cnt, r.cap, r.n = surv.data_ptr(), (sc16 if fused else rc16).data_ptr(), cap, nq
        r.cand_val, r.cand_id, r.cand_cnt, r.cand_cap = cv.data_ptr(), ci.data_ptr(), cc.data_ptr(), ccap
        if fused:
            r.qmeta, r.k, r.out_cnt = qmeta.data_ptr(), k, rc16.data_ptr()
        _lib.call_desc("gnnlm_ivfpq_rescore", r)
        self.stats.add("survivors", lambda sc=sc: sc.sum().double())
        self.stats.add("candidates", lambda cc=cc: cc.sum().double())
        if getattr(self, "keep_candidates", False):
            self.last_candidates = (cv, ci, cc, tau)
        ops.topk_merge(cv, bv, bi, ids=ci, largest=True, init=True, row_ncols=cc.clamp(max=ccap))
        over = sc                                                              # every candidate is a survivor: sc >= cc
        if ccap < cap:                                                         # more candidates than columns: an overflow like any other
            over = torch.where(cc > ccap, torch.full_like(sc, self.COLUMNS), over)
        if S > 1:                                                              # the sampled threshold's proof: k candidates above it (or every key there is)
            avail = (self.list_off[1:] - self.list_off[:-1])[pi.clamp(min=0)].masked_fill(pi < 0, 0).sum(1)
            short = cc.to(torch.int64) < avail.clamp(max=k)
            self.stats.add("underflow", lambda short=short: short.sum().double())
            over = torch.where(short, torch.full_like(sc, self.UNDERFLOW), over)
        return over

    def check(self):
        """Kept for callers of earlier versions: ``search_device`` itself re-searches queries whose survivors did not fit."""
        return None

    def search(self, queries, k):
        d, i = self.search_device(torch.as_tensor(np.asarray(queries)), k)
        return d.cpu().numpy(), i.cpu().numpy()
