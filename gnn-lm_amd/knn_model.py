"""``KNNModel`` -- mirror of ``knn/knn_model.py:25-217``: kNN-LM probabilities from a datastore.

Same constructor arguments, attributes and ``get_knns`` / ``get_knn_prob`` contracts.  Differences:
  * faiss is optional.  ``index_file`` is opened with faiss when it is importable; otherwise pass
    ``index=`` (any object with ``search(queries_f32, k) -> (dists, ids)``, faiss's contract) or let the
    model build an :class:`ExactIndex` (exact search on the GPU; NOT the reference's approximate
    ``OPQ64_1024,IVF4096,PQ64`` index -- ANN results / ADC distances are parity-unpinned).
  * the label table lives in HBM and the mask / distance-softmax / target-match / recall arithmetic
    (knn_model.py:192-217) is one HIP kernel; ``interpolate`` additionally fuses
    ``log(p + 1e-10)`` and the log-space mix of ``sequence_scorer.py:55-68,121``.
  * "cosine" behaviour is keyed off the index file NAME exactly like the reference (:172,181).
"""
import logging
import os
from time import time

import numpy as np
import torch

from . import _lib, ops
from .data_store import DataStore

LOGGING = logging.getLogger(__name__)


class ExactIndex:
    """Exact top-k search on the device with faiss's ``search`` contract (IP: larger = closer, returned
    descending; L2: squared distances ascending; missing results padded with id -1).

    Scales with the store: the keys stay in HBM in their stored dtype (fp16 ``keys.npy`` is read in place, no f32
    copy of the table), the search walks them in chunks -- chunk -> f32 (one conversion pass, ~2 % of the chunk's
    GEMM) -> scores of the chunk on the f32 MFMA GEMM -> ``gnnlm_topk_merge`` folds them into the running top-k.
    Neither the [n, N] score matrix nor a sort of it ever exists.  Cosine: the per-key 1/|key| is a column scale of
    the merge (index_builder.py:90-95,118 normalises the keys it adds)."""

    def __init__(self, keys, metric="ip", cosine=False, device="cuda", chunk_rows=65536, score_bytes=256 << 20, id_map=None):
        self.metric, self.cosine, self.device = metric, cosine, torch.device(device)
        # id_map: row -> external id (faiss IndexIDMap: add_with_ids, index_builder.py:99); None = the row number
        self.id_map = None if id_map is None else torch.as_tensor(np.asarray(id_map), dtype=torch.int64).to(self.device)
        self.chunk_rows, self.score_bytes = chunk_rows, score_bytes
        if isinstance(keys, torch.Tensor):
            self.keys = keys.to(self.device).contiguous()
        else:                                               # numpy array / memmap: upload piecewise, dtype kept
            n, d = keys.shape
            dt = torch.float16 if keys.dtype == np.float16 else torch.float32
            self.keys = torch.empty(n, d, dtype=dt, device=self.device)
            step = max(1, (256 << 20) // max(1, d * keys.dtype.itemsize))
            for r0 in range(0, n, step):
                self.keys[r0:r0 + step] = torch.from_numpy(np.ascontiguousarray(keys[r0:r0 + step])).to(self.device)
        self.ntotal, self.d = self.keys.shape
        self.col_scale = self.col_bias = None
        if cosine or metric == "l2":                        # one pass over the keys: |key|^2
            n2 = torch.empty(self.ntotal, device=self.device, dtype=torch.float32)
            for r0 in range(0, self.ntotal, chunk_rows):
                n2[r0:r0 + chunk_rows] = (self._chunk(r0) ** 2).sum(-1)
            if cosine:
                self.col_scale = n2.rsqrt()
            if metric == "l2":
                self.col_bias = n2 * self.col_scale ** 2 if cosine else n2

    def _chunk(self, r0):
        kc = self.keys[r0:r0 + self.chunk_rows]
        return ops.half_to_float(kc) if kc.dtype == torch.float16 else kc

    def search_device(self, q, k):
        q = q.to(self.device, torch.float32).contiguous()
        n = q.shape[0]
        val = torch.empty(n, k, device=self.device, dtype=torch.float32)
        idx = torch.empty(n, k, device=self.device, dtype=torch.int64)
        ip = self.metric == "ip"
        qb = max(1, min(n, self.score_bytes // (4 * min(self.chunk_rows, max(self.ntotal, 1)))))
        for q0 in range(0, n, qb):
            qs, vs, ids = q[q0:q0 + qb], val[q0:q0 + qb], idx[q0:q0 + qb]
            for r0 in range(0, max(self.ntotal, 1), self.chunk_rows):
                kc = self._chunk(r0)
                if kc.shape[0] == 0:
                    s = torch.empty(qs.shape[0], 0, device=self.device, dtype=torch.float32)
                else:
                    s = ops.gemm_nt(qs, kc)
                sl = slice(r0, r0 + kc.shape[0])
                ops.topk_merge(s, vs, ids, col0=r0,
                               col_scale=None if self.col_scale is None else self.col_scale[sl],
                               col_bias=None if self.col_bias is None else self.col_bias[sl],
                               alpha=1.0 if ip else -2.0, largest=ip, init=(r0 == 0))
        if not ip:
            val += (q ** 2).sum(-1, keepdim=True)           # |q|^2 + |k|^2 - 2 q.k
        if self.id_map is not None and self.ntotal:
            idx = torch.where(idx >= 0, self.id_map[idx.clamp_min(0)], idx)
        return val, idx

    def search(self, queries, k):
        d, i = self.search_device(torch.as_tensor(np.asarray(queries)), k)
        return d.cpu().numpy(), i.cpu().numpy()


class KNNModel(object):
    def __init__(self, index_file, dstore_dir, probe: int = 32, no_load_keys: bool = False,
                 metric_type: str = "do_not_recomp_ip", sim_func: str = None, k: int = 1024, cuda: int = -1,
                 use_memory=False, efsearch=8, index=None, device=None):
        self.index_file, self.dstore_dir = index_file or "", dstore_dir
        self.probe, self.efsearch = probe, efsearch
        self.no_load_keys, self.use_memory = no_load_keys, use_memory
        if not os.path.exists(dstore_dir):
            raise ValueError(f"Dstore path not found: {dstore_dir}")                     # knn_model.py:74
        t = time()
        self.data_store = DataStore.from_pretrained(dstore_dir=dstore_dir, use_memory=use_memory,
                                                    no_load_keys=no_load_keys)
        LOGGING.info(f"Reading datastore took {time() - t} s")
        self.dstore_size, self.hidden_size = self.data_store.dstore_size, self.data_store.hidden_size
        self.vocab_size, self.dstore_fp16 = self.data_store.vocab_size, self.data_store.dstore_fp16
        self.vals = self.data_store.vals
        if not no_load_keys:
            self.keys = self.data_store.keys
        self.k, self.metric_type, self.sim_func, self.cuda = k, metric_type, sim_func, cuda
        assert self.metric_type in ["do_not_recomp_l2", "do_not_recomp_ip", "l2", "ip"]
        self.device = torch.device(device if device is not None else "cuda")
        # `ip` / `l2` recompute similarities from the keys: the table goes to HBM when it fits (and is smaller than this bound),
        # else its rows are gathered from the host per query block (`host_gather_bytes` of float32 rows per block)
        self.max_hbm_key_bytes, self.host_gather_bytes = float("inf"), 1 << 30
        self.index = index if index is not None else self.setup_faiss()
        self._vals_dev = None
        # an IVF-PQ index searched on the device carries the labels next to the key ids (one payload per key): the search
        # hands `self.vals[knns]` (knn_model.py:198) back with the neighbours and the per-query label gather disappears
        if getattr(self.index, "attach_vals", None) is not None and not self.index.has_vals and self.data_store.val_size == 1:
            self.index.attach_vals(self.vals_device())

    @property
    def cosine(self):
        return "cosine" in self.index_file

    def setup_faiss(self):
        """The index behind ``get_knns`` (knn_model.py:59-64,78-82), in this order: an IVF-PQ index in this package's
        own format next to ``index_file`` (``<index_file>.gnnlm.npz``, written by ``python -m gnnlm_amd.run_index_build``:
        searched on the GPU, the hot-path choice at datastore scale); ``index_file`` itself if it is a faiss
        ``[OPQ,]IVF,PQ`` inner-product index (read by faiss_io, searched on the GPU); a faiss index if faiss is importable; else an
        exact index over the keys resident in HBM (chunked search, fine up to a few 10^7 keys)."""
        own = self.index_file if self.index_file.endswith(".gnnlm.npz") else self.index_file + ".gnnlm.npz"
        if os.path.exists(own):
            from .ivfpq import IVFPQIndex
            LOGGING.info("IVF-PQ index %s searched on %s", own, self.device)
            return IVFPQIndex.load(own, device=self.device, nprobe=self.probe)
        if os.path.isfile(self.index_file):
            from . import faiss_io
            if faiss_io.sniff(self.index_file) in ("IxPT", "IwPQ"):
                # the reference's own faiss file (knn/index_builder.py): read without faiss, searched on the GPU
                from .ivfpq import IVFPQIndex
                try:
                    index = IVFPQIndex.from_faiss_file(self.index_file, device=self.device, cosine=self.cosine, nprobe=self.probe)
                    LOGGING.info("faiss IVF-PQ file %s searched on %s", self.index_file, self.device)
                    return index
                except ValueError as e:
                    LOGGING.warning("%s", e)
            elif faiss_io.sniff(self.index_file) in ("IxMp", "IxM2", "IxFI", "IxF2", "IxFl"):
                # `IDMap,,Flat`: what index_builder.py:49-53 builds for small datastores -- exact search over the file's vectors
                z = faiss_io.read_flat_index(self.index_file)
                LOGGING.info("faiss Flat file %s (%d vectors) searched on %s", self.index_file, z["xb"].shape[0], self.device)
                return ExactIndex(torch.from_numpy(np.ascontiguousarray(z["xb"])), z["metric"], False, self.device, id_map=z["ids"])
        try:
            import faiss
        except ImportError:
            faiss = None
        if faiss is not None and os.path.exists(self.index_file):
            index = faiss.read_index(self.index_file, faiss.IO_FLAG_ONDISK_SAME_DIR)
            try:
                faiss.ParameterSpace().set_index_parameter(index, "nprobe", self.probe)
                faiss.ParameterSpace().set_index_parameter(index, "quantizer_efSearch", self.efsearch)
            except Exception:
                LOGGING.warning(f"faiss index {self.index_file} does not have parameter nprobe or efSearch")
            return index
        if self.no_load_keys:
            raise ValueError("faiss is not installed and the keys were not loaded: pass index=... "
                             "(an object with faiss's search contract) or no_load_keys=False")
        LOGGING.warning("faiss not available: exact search over %d keys on the GPU", self.dstore_size)
        base = "l2" if self.metric_type.endswith("l2") else "ip"
        return ExactIndex(self.keys, base, self.cosine, self.device)

    def _keys_device(self):
        if isinstance(getattr(self.index, "keys", None), torch.Tensor):
            return self.index.keys
        if isinstance(self.keys, torch.Tensor):
            return self.keys
        need = self.dstore_size * self.hidden_size * (2 if self.dstore_fp16 else 4)
        free = torch.cuda.mem_get_info(self.device)[0]
        if need > min(0.9 * free, self.max_hbm_key_bytes):
            return None                               # the key table stays where the reference keeps it: _sims gathers rows from the host
        return self.data_store.keys_to_device(self.device)

    def vals_device(self):
        if self._vals_dev is None:
            self._vals_dev = self.data_store.vals_to_device(self.device)
        return self._vals_dev

    # ------------------------------------------------------------------------------------------
    def get_knns(self, queries, k: int = 0):
        """-> (dists [num,k] f32, knns [num,k] i64) as numpy arrays (knn_model.py:87-101)."""
        k = k or self.k
        if isinstance(queries, torch.Tensor):
            queries = queries.detach().cpu().float().data.numpy()
        return self.index.search(queries.astype(np.float32), k)

    def _search(self, knn_queries, k):
        """-> (dists, knns, knn_vals or None): knn_vals = vals[knns] when the index delivers the labels itself."""
        if getattr(self.index, "has_vals", False):
            return self.index.search_device(knn_queries, k, return_vals=True)
        if hasattr(self.index, "search_device"):
            return tuple(self.index.search_device(knn_queries, k)) + (None,)
        d, i = self.get_knns(knn_queries, k)
        return torch.from_numpy(d).to(self.device), torch.from_numpy(i).to(self.device), None

    def _sims(self, dists, knns, queries):
        """sim_func dispatch of knn_model.py:137-177 (device tensors)."""
        fn = self.metric_type
        if fn == "do_not_recomp_l2":
            return -1 * dists
        if fn == "do_not_recomp_ip":
            return dists
        # the recomputed similarities gather key rows: from HBM (the exact index's table, or the store's keys uploaded
        # once -- the reference's per-query np.memmap gather on the host, :163,170, is what this replaces); numpy's
        # negative-index wrap of the -1 padding is kept (row N - 1)
        if fn not in ("l2", "ip"):
            raise ValueError("Invalid knn similarity function!")
        keys_dev = self._keys_device()

        def sims_of(vecs, q):
            if fn == "l2":
                return -1 * torch.sum((q[:, None, :] - vecs) ** 2, dim=2)
            if self.cosine:
                vecs = vecs / (vecs ** 2).sum(-1, keepdims=True).sqrt()
            return (vecs * q[:, None, :]).sum(dim=-1)

        if keys_dev is not None:
            idx = torch.where(knns < 0, knns + keys_dev.shape[0], knns)
            return sims_of(keys_dev[idx].float(), queries)
        # A key table larger than the free HBM (WikiText-103 train: 211 GB of fp16 keys next to everything else): the k rows of
        # every query are gathered from the host table exactly as the reference does (`self.keys[knns]` on the np.memmap,
        # :163,170) -- in blocks of queries, through pinned memory -- and the arithmetic runs on the device.  Slow by
        # construction (n k random 2-KB host reads: the recipes use do_not_recomp_ip for that reason); same numbers.
        n, k = knns.shape
        host_idx = torch.where(knns < 0, knns + self.dstore_size, knns).cpu().numpy()
        rows_per_block = max(1, self.host_gather_bytes // max(1, k * self.hidden_size * 4))
        out = torch.empty(n, k, device=queries.device, dtype=torch.float32)
        for r0 in range(0, n, rows_per_block):
            blk = host_idx[r0:r0 + rows_per_block]
            vecs = torch.from_numpy(np.ascontiguousarray(self.keys[blk.reshape(-1)])).to(queries.device).float()
            out[r0:r0 + blk.shape[0]] = sims_of(vecs.reshape(blk.shape[0], k, self.hidden_size), queries[r0:r0 + blk.shape[0]])
        return out

    def search_sims(self, queries, k=0, with_vals=False):
        """queries [n, d] (device) -> (sims [n,k] f32, knns [n,k] i64[, knn_vals [n,k] i32 or None]), before the -1 masking."""
        k = k or self.k
        q = queries.float()
        if self.cosine:                                                                 # :181-184
            q = q / (q ** 2).sum(-1, keepdims=True).sqrt()
        dists, knns, kvals = self._search(q, k)
        out = (self._sims(dists, knns, q).contiguous(), knns.contiguous())
        return out + (None if kvals is None else kvals.contiguous(),) if with_vals else out

    def get_knn_prob(self, queries, k: int = 0, output_size: int = None, return_knn: bool = False, t: float = 1.0,
                     targets: torch.Tensor = None, return_recall: bool = False):
        """Return shapes as documented at knn_model.py:122-128."""
        assert self.data_store.val_size == 1, "make sure self.data_store.val_size == 1 (which is labels)"
        if not (output_size or self.vocab_size):
            raise ValueError("DataStore.info does not have vocab_size, please set output_size manually")
        if not queries.is_cuda:
            raise _lib.GnnlmError("KNNModel.get_knn_prob runs on the GPU; there is no CPU fallback")
        sims, knns, kvals = self.search_sims(queries, k, with_vals=True)
        vals = self.vals_device()
        if targets is None:                     # dense [batch, V] variant: not on the eval path (torch ops)
            masked = sims.masked_fill(knns == -1, -1e10)
            probs = torch.softmax(masked / t, dim=-1)
            knn_vals = vals[knns].long()
            out = torch.zeros(sims.shape[0], output_size or self.vocab_size, device=sims.device)
            out.scatter_add_(1, knn_vals, probs)
            return (out, masked, knns.cpu().numpy()) if return_knn else out
        n = sims.shape[0]
        zeros = torch.zeros(n, device=sims.device, dtype=torch.float32)
        _, p, recall = ops.knn_interp(zeros, sims, knns, targets.to(sims.device).long().contiguous(), t, 0.5,
                                      vals=vals if kvals is None else None, n_store=self.dstore_size, knn_vals=kvals)
        return (p, recall) if return_recall else p

    def interpolate_begin(self, queries, k=0):
        """Start the search for ``interpolate`` and return a handle for ``interpolate_finish``: the scorer enqueues the language
        model's softmax in between, so the device has work while the host waits for the search's survivor count (an index without
        ``search_begin`` is searched here and now)."""
        k = k or self.k
        q = queries.float()
        if self.cosine:                                                                 # :181-184
            q = q / (q ** 2).sum(-1, keepdims=True).sqrt()
        if getattr(self.index, "has_vals", False) and hasattr(self.index, "search_begin") and self.metric_type.startswith("do_not_recomp"):
            return ("pending", q, self.index.search_begin(q.contiguous(), k, return_vals=True))
        dists, knns, kvals = self._search(q, k)
        return ("done", q, (dists, knns, kvals))

    def interpolate_finish(self, handle, targets, lm_logp, t, lmbda):
        kind, q, h = handle
        dists, knns, kvals = h.result() if kind == "pending" else h
        sims = self._sims(dists, knns, q).contiguous()
        return ops.knn_interp(lm_logp.contiguous(), sims, knns.contiguous(), targets.long().contiguous(), t, lmbda,
                              vals=self.vals_device() if kvals is None else None, n_store=self.dstore_size,
                              knn_vals=None if kvals is None else kvals.contiguous())

    def interpolate(self, queries, targets, lm_logp, t, lmbda, k=0):
        """Fused hot-path form: search -> (interpolated log-prob [n], p_knn [n], recall [n])."""
        sims, knns, kvals = self.search_sims(queries, k, with_vals=True)
        return ops.knn_interp(lm_logp.contiguous(), sims, knns, targets.long().contiguous(), t, lmbda,
                              vals=self.vals_device() if kvals is None else None, n_store=self.dstore_size, knn_vals=kvals)
