#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Run in the authoring container only (needs /root/reference; the GPU box has neither the
reference nor this need):

    python tests/golden/make_golden.py

What is executed from /root/reference (nothing of it is copied into this repo):
  * knn/data_store.py                    imported as is
  * knn/pq_wrapper.py, knn/knn_model.py  imported with an EMPTY ``faiss`` module pre-seeded in
                                         sys.modules (faiss is only touched in constructors and in
                                         ``index.search``; objects are built with __new__ + buffers
                                         and a brute-force stand-in for ``index``)
  * fairseq/models/hgt.py                exec'd up to its ``__main__`` guard under the pure-torch
                                         DGL stand-in defined below (``dgl`` is absent from the image)
  * fairseq/data/token_block_dataset.py  the methods new_build_graph / build_ntgt_edges /
                                         auto_regressive_edges, extracted with ``ast`` and exec'd
  * fairseq/modules/adaptive_softmax.py, adaptive_input.py   loaded by path (torch-only files)
  * fairseq/sequence_scorer.py           loaded by path with stub ``fairseq.utils.strip_pad`` /
                                         ``fairseq.data.Dictionary``

Outputs are data only: seeded inputs + the arrays the reference code returned.
"""
import ast
import contextlib
import importlib.util
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(1)


# --------------------------------------------------------------------------------------------
# stand-ins for absent third-party packages (this file's own code, documented DGL semantics)
# --------------------------------------------------------------------------------------------
class _Frame(dict):
    pass


class _NodeView:
    def __init__(self, g):
        self.g = g

    def __getitem__(self, nt):
        return types.SimpleNamespace(data=self.g._ndata[nt])


class ShimGraph:
    """Minimal heterograph: what HGTLayer.forward and new_build_graph touch."""

    def __init__(self, edges):
        self._edges = {}
        n = {}
        for (s, e, d), (u, v) in edges.items():
            u = torch.as_tensor(u, dtype=torch.int64)
            v = torch.as_tensor(v, dtype=torch.int64)
            self._edges[(s, e, d)] = (u, v)
            n[s] = max(n.get(s, 0), int(u.max()) + 1 if u.numel() else 0)
            n[d] = max(n.get(d, 0), int(v.max()) + 1 if v.numel() else 0)
        self._n = n
        self._ndata = {nt: _Frame() for nt in n}
        self._edata = {et: _Frame() for et in self._edges}
        self.nodes = _NodeView(self)

    ntypes = property(lambda self: list(self._n.keys()))
    canonical_etypes = property(lambda self: list(self._edges.keys()))

    def num_nodes(self, nt):
        return self._n[nt]

    @contextlib.contextmanager
    def local_scope(self):
        nd = {k: _Frame(v) for k, v in self._ndata.items()}
        ed = {k: _Frame(v) for k, v in self._edata.items()}
        try:
            yield
        finally:
            self._ndata.clear(); self._ndata.update(nd)
            self._edata.clear(); self._edata.update(ed)

    def __getitem__(self, key):
        return _SubGraph(self, tuple(key))

    def multi_update_all(self, etype_dict, cross_reducer):
        assert cross_reducer == "mean"
        per_dst, out_name = {}, {}
        for (s, e, d), (mfunc, rfunc) in etype_dict.items():
            u, v = self._edges[(s, e, d)]
            kind, sf, ef, mf = mfunc
            assert kind == "u_mul_e"
            m = self._ndata[s][sf][u] * self._edata[(s, e, d)][ef]
            rk, rmf, of = rfunc
            assert rk == "sum" and rmf == mf
            red = torch.zeros((self._n[d],) + m.shape[1:], dtype=m.dtype)
            red.index_add_(0, v, m)                      # zero in-degree -> 0
            per_dst.setdefault(d, []).append(red)
            out_name[d] = of
        for d, lst in per_dst.items():
            self._ndata[d][out_name[d]] = torch.stack(lst, 0).mean(0)


class _SubGraph:
    def __init__(self, g, et):
        self.g, self.et = g, et
        self.srcdata = g._ndata[et[0]]
        self.dstdata = g._ndata[et[2]]
        self.edata = g._edata[et]

    def apply_edges(self, func):
        kind, a, b, out = func
        assert kind == "v_dot_u"
        u, v = self.g._edges[self.et]
        self.edata[out] = (self.dstdata[a][v] * self.srcdata[b][u]).sum(-1, keepdim=True)


def _edge_softmax(sub, score, norm_by="dst"):
    assert norm_by == "dst"
    u, v = sub.g._edges[sub.et]
    n = sub.g._n[sub.et[2]]
    out = torch.empty_like(score)
    for node in range(n):
        m = v == node
        if m.any():
            out[m] = torch.softmax(score[m], dim=0)
    return out


def install_stubs():
    faiss = types.ModuleType("faiss")
    sys.modules["faiss"] = faiss
    dgl = types.ModuleType("dgl")
    dgl.DGLHeteroGraph = ShimGraph
    dgl.DGLGraph = ShimGraph
    dgl.heterograph = lambda edges: ShimGraph(edges)
    fn = types.ModuleType("dgl.function")
    fn.v_dot_u = lambda a, b, out: ("v_dot_u", a, b, out)
    fn.u_mul_e = lambda a, b, out: ("u_mul_e", a, b, out)
    fn.sum = lambda m, out: ("sum", m, out)
    ops = types.ModuleType("dgl.ops")
    ops.edge_softmax = _edge_softmax
    dgl.function, dgl.ops = fn, ops
    sys.modules.update({"dgl": dgl, "dgl.function": fn, "dgl.ops": ops})
    fs = types.ModuleType("fairseq")
    fs.__path__ = []
    inc = types.ModuleType("fairseq.incremental_decoding_utils")
    inc.with_incremental_state = lambda cls: cls
    futils = types.ModuleType("fairseq.utils")
    futils.strip_pad = lambda tensor, pad: tensor[tensor.ne(pad)]     # fairseq/utils.py:190-191
    fdata = types.ModuleType("fairseq.data")
    fdata.Dictionary = type("Dictionary", (), {})
    fs.utils, fs.data = futils, fdata
    sys.modules.update({"fairseq": fs, "fairseq.incremental_decoding_utils": inc,
                        "fairseq.utils": futils, "fairseq.data": fdata})
    sys.path.insert(0, REF)


def load_by_path(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_hgt():
    src = open(os.path.join(REF, "fairseq/models/hgt.py")).read()
    src = src[: src.index("if __name__ == '__main__':")]
    ns = {"__name__": "ref_hgt"}
    exec(compile(src, "ref:fairseq/models/hgt.py", "exec"), ns)
    return ns


def load_graph_builder():
    """class with the reference's new_build_graph / build_ntgt_edges / auto_regressive_edges."""
    src = open(os.path.join(REF, "fairseq/data/token_block_dataset.py")).read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "GraphTokenBlockDataset")
    keep = {"new_build_graph", "build_ntgt_edges", "auto_regressive_edges"}
    cls.body = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in keep]
    cls.bases = []
    mod = ast.Module(body=[cls], type_ignores=[])
    ast.fix_missing_locations(mod)
    import dgl
    from functools import lru_cache
    from typing import Dict, List, Tuple
    ns = {"np": np, "torch": torch, "dgl": dgl, "lru_cache": lru_cache,
          "Dict": Dict, "List": List, "Tuple": Tuple}
    exec(compile(mod, "ref:fairseq/data/token_block_dataset.py", "exec"), ns)
    return ns["GraphTokenBlockDataset"]


class _Len:
    """token_block_dataset.py:384 calls len() on ``neighbor_offsets.shape[0]`` (a reference bug:
    TypeError on an int).  Handing it an object whose len() is the neighbour-corpus length lets the
    rest of the reference function run unmodified with the intended bound."""

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


def ref_build_graph(GB, nb, tgt_offsets, codes, vals, left, right, n_store, invalid_neighbor_context=0):
    self = GB.__new__(GB)
    self.invalid_neighbor_context = invalid_neighbor_context
    self.quant_neighbor_feats = codes
    self.neighbor_tokens = vals.reshape(-1, 1)
    self.left_neighbor_context, self.right_neighbor_context = left, right
    self.neighbor_offsets = types.SimpleNamespace(shape=(_Len(n_store), nb.shape[1]))
    self.max_intra_context = 0
    T = nb.shape[0]
    g = self.new_build_graph(torch.zeros(T, dtype=torch.long), tgt_offsets, nb,
                             torch.zeros(T, dtype=torch.long))
    return g


# --------------------------------------------------------------------------------------------
def gen_datastore():
    from knn.data_store import DataStore
    rs = np.random.RandomState(11)
    out = {}
    for name, fp16, vocab, val_size in [("fp16_i16", True, 205, 1), ("fp16_i32", True, 40000, 1),
                                        ("fp32_i32", False, 205, 1), ("fp16_v2", True, 40000, 2)]:
        N, d = 37, 12
        keys = rs.randn(N, d).astype(np.float16 if fp16 else np.float32)
        vdt = np.int16 if (fp16 and vocab < 2 ** 15) else np.int32
        vals = rs.randint(0, vocab, size=(N, val_size)).astype(vdt)
        with tempfile.TemporaryDirectory() as td:
            keys.tofile(os.path.join(td, "keys.npy"))
            vals.tofile(os.path.join(td, "vals.npy"))
            info = {"dstore_size": N, "hidden_size": d, "vocab_size": vocab, "dstore_fp16": fp16,
                    "val_size": val_size}
            json.dump(info, open(os.path.join(td, "info.json"), "w"))
            for um in (False, True):
                ds = DataStore.from_pretrained(td, use_memory=um)
                assert ds.info == info
                out[f"{name}.keys"] = np.array(ds.keys)
                out[f"{name}.vals"] = np.array(ds.vals)
            ds = DataStore.from_pretrained(td, no_load_keys=True)
            assert not hasattr(ds, "keys")
        out[f"{name}.info"] = np.frombuffer(json.dumps(info, sort_keys=True).encode(), dtype=np.uint8)
        out[f"{name}.keys_raw"] = np.frombuffer(keys.tobytes(), dtype=np.uint8)
        out[f"{name}.vals_raw"] = np.frombuffer(vals.tobytes(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "datastore.npz"), **out)


def make_ref_codec(pqw, cen, A, b, metric="ip"):
    """TorchPQCodec without faiss: replay NumpyPQCodec.__init__'s table code path by hand
    (pq_wrapper.py:37-49 are plain numpy on ``cen``), then the buffer registration (:93-102)."""
    c = pqw.TorchPQCodec.__new__(pqw.TorchPQCodec)
    torch.nn.Module.__init__(c)
    c.metric = metric
    c.pre = (A, b) if A is not None else None
    c.centroids = cen
    c.norm2_centroids = (cen ** 2).sum(axis=2)
    if metric == "l2":
        c.sdc_table = -np.sqrt(((cen[:, :, None] - cen[:, None]) ** 2).sum(3))
    else:
        c.sdc_table = np.matmul(cen, cen.transpose(0, 2, 1))
    if c.pre:
        c.pre_torch = True
        c.register_buffer("A", torch.from_numpy(A))
        c.register_buffer("b", torch.from_numpy(b))
    else:
        c.pre_torch = False
    c.register_buffer("centroids_torch", torch.from_numpy(c.centroids))
    c.register_buffer("norm2_centroids_torch", torch.from_numpy(c.norm2_centroids))
    c.register_buffer("sdc_table_torch", torch.from_numpy(c.sdc_table))
    return c


def gen_pq():
    import knn.pq_wrapper as pqw
    rs = np.random.RandomState(5)
    out = {}
    cases = {
        # name: (M, dsub, d_in, has_pre, has_b)
        "sq_pre": (8, 4, 32, True, True),
        "sq_pre_nob": (8, 4, 32, True, False),
        "rect_pre": (4, 4, 32, True, True),          # OPQ d_out = 16 != d_in = 32
        "nopre": (8, 4, 32, False, False),
    }
    for name, (M, dsub, d_in, pre, has_b) in cases.items():
        d_out = M * dsub
        cen = (rs.randn(M, 256, dsub) * 0.5).astype(np.float32)
        A = (rs.randn(d_out, d_in) / np.sqrt(d_in)).astype(np.float32) if pre else None
        b = ((rs.randn(d_out) * 0.1).astype(np.float32) if has_b else np.zeros(0, np.float32)) if pre else None
        x = rs.randn(64, d_in if pre else d_out).astype(np.float32)
        for metric in ("ip", "l2"):
            codec = make_ref_codec(pqw, cen, A, b, metric)
            codes = codec.encode(torch.from_numpy(x.copy()))
            dec = codec.decode(codes)
            np_dec = pqw.NumpyPQCodec.decode(codec, codes.numpy())
            np_enc = pqw.NumpyPQCodec.encode(codec, x.copy())
            assert np.array_equal(np_enc, codes.numpy())
            assert np.allclose(np_dec, dec.numpy(), atol=1e-5)
            sim = codec.compute_sim(codes[:5], codes)
            out[f"{name}.{metric}.sdc_corner"] = codec.sdc_table[:, :8, :8].copy()
            out[f"{name}.{metric}.sdc_rowsum"] = codec.sdc_table.sum(-1)
            out[f"{name}.{metric}.sim"] = sim.numpy()
        out[f"{name}.cen"] = cen
        if pre:
            out[f"{name}.A"], out[f"{name}.b"] = A, b
        out[f"{name}.x"] = x
        out[f"{name}.codes"] = codes.numpy()
        out[f"{name}.decode"] = dec.numpy()
        out[f"{name}.norm2"] = codec.norm2_centroids
    # full-size case: tables regenerated from the seed inside the test, only outputs stored
    rs = np.random.RandomState(77)
    M, dsub, d = 128, 8, 1024
    cen = (rs.randn(M, 256, dsub) * 0.5).astype(np.float32)
    A = (rs.randn(d, d) / np.sqrt(d)).astype(np.float32)
    b = (rs.randn(d) * 0.1).astype(np.float32)
    codes = rs.randint(0, 256, size=(16, M)).astype(np.uint8)
    codec = make_ref_codec(pqw, cen, A, b)
    out["full.codes"] = codes
    out["full.decode"] = codec.decode(torch.from_numpy(codes)).numpy()
    out["full.lookup_checksum"] = np.array([float(np.float64(cen[np.arange(M)[None], codes.astype(np.int64)]).sum())])
    np.savez_compressed(os.path.join(OUT, "pq.npz"), **out)


class BruteIndex:
    """Stand-in for a faiss index: exact search, padding the tail with -1 ids like faiss does
    when fewer than k results exist."""

    def __init__(self, keys, metric, cosine, n_pad=0):
        k = keys.astype(np.float32)
        if cosine:
            k = k / np.sqrt((k ** 2).sum(-1, keepdims=True))
        self.keys, self.metric, self.n_pad = k, metric, n_pad

    def search(self, q, k):
        if self.metric == "ip":
            s = q @ self.keys.T
            ids = np.argsort(-s, axis=1, kind="stable")[:, :k]
        else:
            s = ((q[:, None] - self.keys[None]) ** 2).sum(-1)
            ids = np.argsort(s, axis=1, kind="stable")[:, :k]
        d = np.take_along_axis(s, ids, 1).astype(np.float32)
        ids = ids.astype(np.int64)
        if self.n_pad:
            # a few rows get trailing -1 padding (faiss: dist = -inf for IP / +inf for L2 -> use finite fill)
            for r in range(0, q.shape[0], 3):
                ids[r, -self.n_pad:] = -1
                d[r, -self.n_pad:] = -3.4e38 if self.metric == "ip" else 3.4e38
        return d, ids


def gen_knn():
    import knn.knn_model as km
    from knn.data_store import DataStore
    rs = np.random.RandomState(3)
    N, d, V, n, k = 200, 16, 50, 12, 8
    keys = rs.randn(N, d).astype(np.float16)
    vals = rs.randint(0, V, size=(N, 1)).astype(np.int16)      # data_store.py:50: fp16 store, V < 2**15
    queries = rs.randn(n, d).astype(np.float32)
    targets = rs.randint(0, V, size=n).astype(np.int64)
    out = {"keys": keys, "vals": vals.reshape(-1), "queries": queries, "targets": targets}
    with tempfile.TemporaryDirectory() as td:
        keys.tofile(os.path.join(td, "keys.npy"))
        vals.tofile(os.path.join(td, "vals.npy"))
        json.dump({"dstore_size": N, "hidden_size": d, "vocab_size": V, "dstore_fp16": True,
                   "val_size": 1}, open(os.path.join(td, "info.json"), "w"))
        ds = DataStore.from_pretrained(td, use_memory=True)
        for metric_type in ("do_not_recomp_ip", "do_not_recomp_l2", "ip", "l2"):
            for cosine in (False, True):
                for t in (1.0, 0.01):
                    m = km.KNNModel.__new__(km.KNNModel)
                    m.data_store, m.vals, m.keys = ds, ds.vals, ds.keys
                    m.vocab_size, m.k, m.metric_type = V, k, metric_type
                    m.index_file = "faiss_store.cosine" if cosine else "faiss_store.ip"
                    base = "l2" if "l2" in metric_type else "ip"
                    m.index = BruteIndex(keys, base, cosine, n_pad=2)
                    # make some targets hit: copy the value of a retrieved neighbour
                    q = torch.from_numpy(queries)
                    p, recall = m.get_knn_prob(q, targets=torch.from_numpy(targets), t=t,
                                               return_recall=True)
                    tag = f"{metric_type}.{'cos' if cosine else 'raw'}.t{t}"
                    qn = queries / np.sqrt((queries ** 2).sum(-1, keepdims=True)) if cosine else queries
                    dd, ii = m.index.search(qn.astype(np.float32), k)
                    out[tag + ".dists"], out[tag + ".ids"] = dd, ii
                    out[tag + ".p"], out[tag + ".recall"] = p.numpy(), recall.numpy()
                    dense, sims, knns = m.get_knn_prob(q, t=t, return_knn=True)
                    out[tag + ".dense"], out[tag + ".sims"] = dense.numpy(), sims.numpy()
    # targets guaranteed to be retrieved for half of the rows (non-trivial recall)
    np.savez_compressed(os.path.join(OUT, "knn.npz"), **out)


def gen_combine():
    src = open(os.path.join(REF, "fairseq/sequence_scorer.py")).read()
    fn = next(n for n in ast.walk(ast.parse(src))
              if isinstance(n, ast.FunctionDef) and n.name == "combine_knn_and_vocab_probs")
    mod = ast.Module(body=[fn], type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = {"torch": torch, "np": np}
    exec(compile(mod, "ref:fairseq/sequence_scorer.py", "exec"), ns)
    rs = np.random.RandomState(9)
    lp = np.log(rs.uniform(1e-6, 1, size=(3, 40))).astype(np.float32)
    pk = rs.uniform(0, 1, size=(3, 40)).astype(np.float32)
    pk[0, :10] = 0.0
    out = {"lm_logp": lp, "p_knn": pk}
    for lm in (0.1, 0.15, 0.2, 0.25):
        knn_logp = torch.log(torch.from_numpy(pk) + 1e-10)          # sequence_scorer.py:110,121
        out[f"mix.{lm}"] = ns["combine_knn_and_vocab_probs"](knn_logp, torch.from_numpy(lp), lm).numpy()
    np.savez_compressed(os.path.join(OUT, "combine.npz"), **out)


def gen_graph_and_hgt():
    GB = load_graph_builder()
    H = load_hgt()
    out_g, out_h = {}, {}
    # doctest vectors of build_ntgt_edges (token_block_dataset.py:549-554) + extras
    o2i = {0: 0, 1: 1, 2: 2, 12: 3, 13: 4}
    for ctx, bi in [(3, False), (0, False), (1, True), (1, False), (2, True)]:
        s, t = GB.build_ntgt_edges(o2i, ctx, bi)
        out_g[f"edges.ctx{ctx}.bi{int(bi)}"] = np.array([s, t], dtype=np.int64)
    for L_, mc in [(5, 0), (8, 3), (1, 0)]:
        us, vs = GB.auto_regressive_edges(L_, max_context=mc)
        out_g[f"ar.{L_}.{mc}"] = np.stack([us.numpy(), vs.numpy()])

    rs = np.random.RandomState(21)
    n_store, M, dsub = 60, 4, 4
    d_small = M * dsub
    codes = rs.randint(0, 256, size=(n_store, M)).astype(np.uint8)
    vals = rs.randint(0, 30, size=n_store).astype(np.int32)
    cen = (rs.randn(M, 256, dsub) * 0.5).astype(np.float32)
    out_g.update(codes=codes, vals=vals)
    graphs = {}
    for T, k, l, r in [(8, 4, 0, 0), (8, 4, 1, 1), (8, 4, 2, 2), (6, 3, 2, 0), (6, 3, 0, 2), (5, 2, 3, 1)]:
        nb = rs.randint(0, n_store, size=(T, k)).astype(np.int64)
        nb[1, 1] = -1
        nb[3, :] = -1                         # a token with zero valid neighbours
        nb[0, 0] = 0                          # clipped left context
        nb[2, 0] = n_store - 1                # clipped right context
        nb[4, 0] = 1
        nb[4, 1 % k] = n_store - 2
        g = ref_build_graph(GB, nb, np.zeros(T, np.int64), codes, vals, l, r, n_store)
        tag = f"T{T}k{k}l{l}r{r}"
        out_g[tag + ".nb"] = nb
        for et, (u, v) in g._edges.items():
            out_g[tag + "." + "_".join(et)] = np.stack([u.numpy(), v.numpy()])
        out_g[tag + ".ntgt_codes"] = g.nodes["ntgt"].data["h"].numpy()
        out_g[tag + ".ntgt_labels"] = g.nodes["ntgt"].data["labels"].numpy()
        graphs[tag] = (g, nb, T, k, l, r)
    np.savez_compressed(os.path.join(OUT, "graph.npz"), **out_g)

    # HGT through the reference's own HGT/HGTLayer.forward on reference-built graphs
    import knn.pq_wrapper as pqw
    d = 32
    A = (rs.randn(d_small, d) / np.sqrt(d)).astype(np.float32)       # decode: [n,16] -> [n,32]
    b = (rs.randn(d_small) * 0.1).astype(np.float32)
    codec = make_ref_codec(pqw, cen, A, b)
    out_h.update(cen=cen, A=A, b=b, codes=codes)
    for tag, (g, nb, T, k, l, r) in graphs.items():
        cfgs = [(1, 2), (2, 8), (3, 2)] if (l, r) == (2, 2) else [[(2, 8)], [(1, 2)], [(3, 2)]][(l + 2 * r) % 3]
        for n_layers, n_heads in cfgs:
            torch.manual_seed(100 + n_layers * 10 + n_heads)
            model = H["HGT"](ntype2idx={"tgt": 0, "ntgt": 1}, etype2idx={"intra": 0, "inter": 1},
                             in_dim=d, hidden_dim=d, out_dim=d, n_layers=n_layers, n_heads=n_heads,
                             dropout=0.1, two_stream=False, attn_drop=0.1).eval()
            with torch.no_grad():
                for nm, p in model.named_parameters():      # move every parameter off its init value
                    if "norms" in nm or "relation_pri" in nm or "bias" in nm:
                        p.add_(torch.randn_like(p) * 0.1)
            tgt = torch.from_numpy(rs.randn(T, d).astype(np.float16).astype(np.float32))
            with torch.no_grad(), g.local_scope():
                ntgt = codec.decode(g.nodes["ntgt"].data["h"])          # transformer.py:1043-1045
                g.nodes["ntgt"].data["h"] = ntgt
                etypes = [("tgt", "intra", "tgt"), ("ntgt", "inter", "tgt"), ("ntgt", "intra", "ntgt")]
                # per-layer outputs: run the layers one by one exactly as HGT.forward does (:510-512)
                h = {"tgt": tgt, "ntgt": ntgt}
                per_layer = []
                for i in range(n_layers):
                    h = model.gcs[i](g, h, etypes=etypes, incremental_state=None)
                    per_layer.append(h)
                full = model(g, features={"tgt": tgt}, etypes=etypes)
            assert torch.equal(full["tgt"], per_layer[-1]["tgt"])
            key = f"{tag}.L{n_layers}H{n_heads}"
            out_h[key + ".tgt_in"] = tgt.numpy()
            for i, hh in enumerate(per_layer):
                out_h[key + f".tgt_out{i}"] = hh["tgt"].numpy()
                out_h[key + f".ntgt_out{i}"] = hh["ntgt"].numpy()
            for nm, p in model.state_dict().items():
                out_h[key + ".sd." + nm] = p.numpy()
    np.savez_compressed(os.path.join(OUT, "hgt.npz"), **out_h)


def gen_graph_invalid_context():
    """new_build_graph with ``invalid_neighbor_context`` in {0, 3, 3072} (token_block_dataset.py:360-362; switched on for
    the train split, language_modeling.py:299): neighbours placed on both sides of the |pos - id| < c boundary."""
    GB = load_graph_builder()
    rs = np.random.RandomState(77)
    n_store, M = 9000, 4
    codes = rs.randint(0, 256, size=(n_store, M)).astype(np.uint8)
    vals = rs.randint(0, 30, size=n_store).astype(np.int32)
    out = {"codes": codes, "vals": vals}
    for c in (0, 3, 3072):
        for T, k, l, r, start in [(8, 4, 2, 2, 4000), (6, 3, 1, 0, 0), (7, 5, 0, 2, n_store - 7)]:
            pos = (start + np.arange(T)).astype(np.int64)
            nb = rs.randint(0, n_store, size=(T, k)).astype(np.int64)
            nb[0, 0] = pos[0]                                   # the token itself
            nb[1, 1] = min(n_store - 1, pos[1] + max(c - 1, 0)) # last row inside the window
            nb[2, 0] = max(0, pos[2] - max(c - 1, 0))
            nb[2, 1] = min(n_store - 1, pos[2] + c)             # first row outside it
            nb[3, 2 % k] = max(0, pos[3] - c)
            nb[4, 0] = -1
            nb[5, :] = np.clip(pos[5] + np.arange(k) - 1, 0, n_store - 1)   # a token whose neighbours are all its own context (c >= k)
            g = ref_build_graph(GB, nb, pos, codes, vals, l, r, n_store, invalid_neighbor_context=c)
            tag = f"c{c}.T{T}k{k}l{l}r{r}"
            out[tag + ".nb"], out[tag + ".pos"] = nb, pos
            for et, (u, v) in g._edges.items():
                out[tag + "." + "_".join(et)] = np.stack([u.numpy(), v.numpy()])
            out[tag + ".ntgt_codes"] = g.nodes["ntgt"].data["h"].numpy() if "h" in g.nodes["ntgt"].data else np.zeros((0, M), np.uint8)
            out[tag + ".ntgt_labels"] = g.nodes["ntgt"].data["labels"].numpy()
    np.savez_compressed(os.path.join(OUT, "graph_ctx.npz"), **out)


def gen_graph_no_neighbours():
    """A block in which NO token has a valid neighbour (every id -1): the one state in which an edge type of the graph would
    have no edges at all (('ntgt','inter','tgt') and ('ntgt','intra','ntgt') both need an ntgt node).  The reference never
    gets as far as DGL with it: new_build_graph stacks an empty list of neighbour rows (token_block_dataset.py:407-410) and
    numpy raises.  Recorded as a fact of the reference -- what DGL's cross-type mean does with an edge type without edges is
    therefore not on the path; this build scores such a block from the causal edges alone (tests pin that to the oracle)."""
    GB = load_graph_builder()
    rs = np.random.RandomState(314)
    n_store, M = 60, 4
    codes = rs.randint(0, 256, size=(n_store, M)).astype(np.uint8)
    vals = rs.randint(0, 30, size=n_store).astype(np.int32)
    out = {"codes": codes, "vals": vals}
    for T, k, l, r in [(6, 3, 2, 2), (4, 2, 0, 0)]:
        nb = np.full((T, k), -1, dtype=np.int64)
        raised, where = 0, -1
        try:
            ref_build_graph(GB, nb, np.zeros(T, np.int64), codes, vals, l, r, n_store)
        except ValueError as e:
            import traceback
            raised = 1
            where = [f.lineno for f in traceback.extract_tb(e.__traceback__) if "token_block_dataset" in f.filename][-1]
        tag = f"T{T}k{k}l{l}r{r}"
        out[tag + ".nb"], out[tag + ".reference_raises_value_error"], out[tag + ".raised_at_line"] = nb, np.array([raised]), np.array([where])
        # one valid neighbour is enough for the reference to build the graph again
        nb1 = nb.copy()
        nb1[T - 1, 0] = 7
        g = ref_build_graph(GB, nb1, np.zeros(T, np.int64), codes, vals, l, r, n_store)
        out[tag + ".one.nb"] = nb1
        for et, (u, v) in g._edges.items():
            out[tag + ".one." + "_".join(et)] = np.stack([u.numpy(), v.numpy()])
    np.savez_compressed(os.path.join(OUT, "graph_empty.npz"), **out)


def gen_hgt_adapters():
    """HGT with in_dim != hidden_dim != out_dim: the reference's own forward incl. `F.gelu(adapt_ws[ntype](feat))`
    (hgt.py:505-507) and the output Linear (:513).  Own random stream: the fixtures above do not move."""
    GB = load_graph_builder()
    H = load_hgt()
    import knn.pq_wrapper as pqw
    rs = np.random.RandomState(77)
    n_store, M, dsub = 50, 4, 4
    codes = rs.randint(0, 256, size=(n_store, M)).astype(np.uint8)
    vals = rs.randint(0, 30, size=n_store).astype(np.int32)
    cen = (rs.randn(M, 256, dsub) * 0.5).astype(np.float32)
    out = {"codes": codes, "cen": cen}
    etypes = [("tgt", "intra", "tgt"), ("ntgt", "inter", "tgt"), ("ntgt", "intra", "ntgt")]
    for case, (d_in, d_hid, d_out, L, Hh, T, k, l, r) in enumerate([(24, 32, 24, 2, 4, 6, 3, 1, 1), (16, 32, 16, 1, 2, 5, 4, 2, 0),
                                                                    (32, 16, 32, 3, 2, 7, 2, 0, 2), (32, 32, 8, 2, 8, 6, 3, 2, 2)]):
        A = (rs.randn(M * dsub, d_in) / np.sqrt(d_in)).astype(np.float32)
        b = (rs.randn(M * dsub) * 0.1).astype(np.float32)
        codec = make_ref_codec(pqw, cen, A, b)
        nb = rs.randint(0, n_store, size=(T, k)).astype(np.int64)
        nb[1, 0] = -1
        nb[2, :] = -1
        nb[0, 0], nb[3, 0] = 0, n_store - 1
        g = ref_build_graph(GB, nb, np.zeros(T, np.int64), codes, vals, l, r, n_store)
        torch.manual_seed(500 + case)
        model = H["HGT"](ntype2idx={"tgt": 0, "ntgt": 1}, etype2idx={"intra": 0, "inter": 1}, in_dim=d_in, hidden_dim=d_hid,
                         out_dim=d_out, n_layers=L, n_heads=Hh, dropout=0.1, two_stream=False, attn_drop=0.1).eval()
        with torch.no_grad():
            for nm, p_ in model.named_parameters():
                if "norms" in nm or "relation_pri" in nm or "bias" in nm:
                    p_.add_(torch.randn_like(p_) * 0.1)
        tgt = torch.from_numpy(rs.randn(T, d_in).astype(np.float16).astype(np.float32))
        with torch.no_grad(), g.local_scope():
            g.nodes["ntgt"].data["h"] = codec.decode(g.nodes["ntgt"].data["h"])      # transformer.py:1043-1045
            full = model(g, features={"tgt": tgt}, etypes=etypes)
        key = f"c{case}"
        out[key + ".cfg"] = np.array([d_in, d_hid, d_out, L, Hh, T, k, l, r], dtype=np.int64)
        out[key + ".A"], out[key + ".b"], out[key + ".nb"], out[key + ".tgt_in"] = A, b, nb, tgt.numpy()
        out[key + ".tgt_out"], out[key + ".ntgt_out"] = full["tgt"].numpy(), full["ntgt"].numpy()
        for nm, p_ in model.state_dict().items():
            out[key + ".sd." + nm] = p_.numpy()
    np.savez_compressed(os.path.join(OUT, "hgt_adapt.npz"), **out)


def gen_adaptive_and_scorer():
    asm_mod = load_by_path("ref_adaptive_softmax", "fairseq/modules/adaptive_softmax.py")
    ain_mod = load_by_path("ref_adaptive_input", "fairseq/modules/adaptive_input.py")
    torch.manual_seed(7)
    V, d, cutoff = 24, 16, [8, 16]
    ain = ain_mod.AdaptiveInput(V, 1, d, 4, d, cutoff)
    asm = asm_mod.AdaptiveSoftmax(V, d, cutoff, dropout=0.0, factor=4, adaptive_inputs=ain,
                                  tie_proj=True).eval()
    x = torch.randn(2, 9, d)
    tgt = torch.randint(0, V, (2, 9))
    tgt[0, :3] = torch.tensor([0, 9, 20])
    with torch.no_grad():
        dense = asm.get_log_prob(x, None)
        with_t = asm.get_log_prob(x, tgt)
    out = {"x": x.numpy(), "target": tgt.numpy(), "dense": dense.numpy(),
           "target_logp": with_t.gather(2, tgt.unsqueeze(-1)).squeeze(-1).numpy(),
           "cutoff": np.array(cutoff + [V]),
           "class_proj": asm.head.class_proj.weight.detach().numpy()}
    for i in range(3):
        e, p = ain.weights_for_band(i)
        out[f"emb{i}"], out[f"proj{i}"] = e.detach().numpy(), p.detach().numpy()
    np.savez_compressed(os.path.join(OUT, "adaptive_softmax.npz"), **out)

    # ---- SequenceScorer.generate through the reference code with a scripted model -------------
    import knn.knn_model as km
    sys.modules["knn.knn_model"] = km
    sc = load_by_path("ref_sequence_scorer", "fairseq/sequence_scorer.py")
    rs = np.random.RandomState(13)
    N, k = 120, 6
    keys = rs.randn(N, d).astype(np.float16)
    vals = rs.randint(0, V, size=(N, 1)).astype(np.int16)      # data_store.py:50: fp16 store, V < 2**15
    bsz, T = 2, 9
    feats = torch.randn(bsz, T, d)                     # "gcn_feat" = HGT output, scripted
    inner = torch.randn(T, bsz, d)                     # inner_states[-1] = base-LM features

    class Model(torch.nn.Module):
        def forward(self, src_tokens, src_lengths, graph=None):
            return feats, {"inner_states": [inner], "gcn_feat": feats.transpose(0, 1)}

        def get_normalized_probs(self, net_output, log_probs, sample):
            assert log_probs
            return asm.get_log_prob(net_output[0], target=sample["target"])      # transformer.py:1064-1079

    tgt_dict = types.SimpleNamespace(pad=lambda: 1, eos=lambda: 2)
    target = torch.randint(3, V, (bsz, T))
    target[1, -2:] = 1                                  # right padding on the second row
    sample = {"id": torch.arange(bsz), "nsentences": bsz, "ntokens": bsz * T,
              "net_input": {"src_tokens": torch.zeros(bsz, T, dtype=torch.long),
                            "src_lengths": torch.full((bsz,), T)},
              "target": target, "start_indices": torch.tensor([[0], [2]])}
    res = {"feats": feats.numpy(), "inner": inner.numpy(), "target": target.numpy(),
           "keys": keys, "vals": vals.reshape(-1), "start_indices": np.array([0, 2])}
    with tempfile.TemporaryDirectory() as td:
        keys.tofile(os.path.join(td, "keys.npy"))
        vals.tofile(os.path.join(td, "vals.npy"))
        json.dump({"dstore_size": N, "hidden_size": d, "vocab_size": V, "dstore_fp16": True,
                   "val_size": 1}, open(os.path.join(td, "info.json"), "w"))
        from knn.data_store import DataStore
        ds = DataStore.from_pretrained(td, use_memory=True)
        for keytype in ("gcn_feat", "keytype"):          # 'keytype' = the recipe's typo -> inner_states
            for lmbda, temp in [(0.25, 1.0), (0.1, 0.01), (0.0, 1.0)]:
                m = km.KNNModel.__new__(km.KNNModel)
                m.data_store, m.vals, m.keys = ds, ds.vals, ds.keys
                m.vocab_size, m.k, m.metric_type = V, k, "do_not_recomp_ip"
                m.index_file = "faiss_store.cosine"
                m.index = BruteIndex(keys, "ip", True, n_pad=1)
                args = types.SimpleNamespace(lmbda=lmbda, knn_keytype=keytype)
                scorer = sc.SequenceScorer(tgt_dict, softmax_batch=3072, args=args)
                with torch.no_grad():
                    hyp = scorer.generate([Model()], dict(sample), knn_dstore=m, temperature=temp)
                tag = f"{keytype}.l{lmbda}.t{temp}"
                for i, h in enumerate(hyp):
                    h = h[0]
                    res[f"{tag}.{i}.tokens"] = h["tokens"].numpy()
                    res[f"{tag}.{i}.score"] = np.array(h["score"].item(), dtype=np.float32)
                    res[f"{tag}.{i}.positional_scores"] = h["positional_scores"].numpy()
                    res[f"{tag}.{i}.dstore_keys"] = h["dstore_keys"].numpy()
                    if h["knn_recall"] is not None:
                        res[f"{tag}.{i}.knn_recall"] = h["knn_recall"].numpy()
    np.savez_compressed(os.path.join(OUT, "scorer.npz"), **res)


if __name__ == "__main__":
    install_stubs()
    gen_datastore()
    gen_pq()
    gen_knn()
    gen_combine()
    gen_graph_and_hgt()
    gen_adaptive_and_scorer()
    gen_hgt_adapters()
    gen_graph_invalid_context()
    gen_graph_no_neighbours()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
