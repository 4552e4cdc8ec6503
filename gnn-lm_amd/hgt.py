"""HGT over the implicit token/neighbour graph -- host-side mirror of ``fairseq/models/hgt.py``.

``HGT`` / ``HGTLayer`` keep the reference's constructor signature and parameter names
(``gcs.{i}.{k,q,v,a}_linears.{t}.{weight,bias}``, ``norms``, ``relation_pri/att/msg``, ``skip``;
hgt.py:55-79, :460-492) so a reference checkpoint loads with ``load_state_dict``.  ``forward`` takes a
:class:`NeighborGraph` instead of a DGL heterograph: the graph of the reference
(token_block_dataset.py:338-412) is fully determined by the neighbour-id matrix and the context
sizes, so it is never materialised.

Before the first forward the weights are *prepared* once (float64 on the host, rounded to float32):
  * relation matrices and ``relation_pri / sqrt(d_k)`` are folded into the K / V projections
    (``k' = (h W_k^T + b_k) R``  ==  ``h (W_k^T R) + b_k R``; hgt.py:347-348,355);
  * on the star edges the neighbour-side K / V projections (and, in layer 0, the OPQ rotation of
    ``TorchPQCodec.decode``, pq_wrapper.py:198-202) are absorbed into the query side, see csrc/attn.hip;
all exact algebra -- only the floating-point association changes.
"""
import ctypes
import os
import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib

NTYPE2IDX = {"tgt": 0, "ntgt": 1}
ETYPE2IDX = {"intra": 0, "inter": 1}


@dataclass
class CodeStore:
    """PQ code table (and labels) resident in HBM: rows [row0, row0 + n_local) of an n_store-row store."""
    codes: torch.Tensor                  # uint8 [n_local, M]
    centroids: torch.Tensor              # f32 [M, 256, dsub]
    n_store: int
    row0: int = 0
    vals: Optional[torch.Tensor] = None  # int16/int32 [n_local]
    A: Optional[torch.Tensor] = None     # f32 [M*dsub, d]  OPQ matrix (decode: (x - b) @ A)
    b: Optional[torch.Tensor] = None     # f32 [M*dsub]
    # Range-sharded store with EVERY shard mapped into this process (dist.PeerMappedFetcher.mapped_store): shard g =
    # (codes tensor [rows_g, M], first global row it holds); rows_per_rank = ceil(n_store / world).  The kernels then read
    # a row from its owner's memory directly and `codes` / `row0` only describe the local shard.
    shards: Optional[list] = None
    rows_per_rank: int = 0


def shards_device_ptr(store):
    """Device address of the store's gnnlm_shards_t (built once, kept alive on the store); None for a one-table store."""
    if store is None or not getattr(store, "shards", None):
        return None
    key = (store.rows_per_rank,) + tuple((c.data_ptr() if c is not None else 0, c.shape[0] if c is not None else 0, r0)
                                         for c, r0 in store.shards)
    t = getattr(store, "_shards_dev", None)
    if t is None or getattr(store, "_shards_key", None) != key:          # (re)built when a shard tensor was swapped
        import ctypes
        assert len(store.shards) <= 16 and store.rows_per_rank > 0
        sh = _lib.gnnlm_shards_t()
        sh.n, sh.rows_per_rank = len(store.shards), store.rows_per_rank
        for g, (c, r0) in enumerate(store.shards):
            sh.base[g] = c.data_ptr() if c is not None and c.numel() else None
            sh.row0[g], sh.rows[g] = r0, (c.shape[0] if c is not None else 0)
        dev = next(c.device for c, _ in store.shards if c is not None)
        t = torch.frombuffer(bytearray(ctypes.string_at(ctypes.byref(sh), ctypes.sizeof(sh))), dtype=torch.uint8).to(dev)
        store._shards_dev, store._shards_key = t, key
    return t.data_ptr()


@dataclass
class NeighborGraph:
    """What replaces the DGL graph: ``ids[b*T + i, j]`` = datastore row of neighbour j of token i."""
    ids: torch.Tensor                    # int64 [n_blocks*T, kg], -1 = none
    n_blocks: int
    T: int
    left: int
    right: int
    store: CodeStore
    tgt_h: Optional[torch.Tensor] = None            # f32 [n_blocks*T, d]  (graph.nodes['tgt'].data['h'])
    fetched_codes: Optional[torch.Tensor] = None    # sharded store: codes of every slot, already fetched
    fetched_valid: Optional[torch.Tensor] = None
    fetched_centres_only: bool = False
    fetched_index: Optional[torch.Tensor] = None    # int32: slot s lives in row fetched_index[s] of fetched_codes
    max_intra_context: int = 0
    # sharded store: the object that fetches code rows from their owners (dist.ShardedFetcher / PeerMappedFetcher interface:
    # fetch_codes(ids, left, right, centres_only), fetch_groups(centres, left, right, counters)).  HGT.forward then fetches itself
    # -- for a multi-layer model AFTER merging equal context groups, so every distinct centre row is requested once.
    fetcher: Optional[object] = None

    @property
    def kg(self):
        return self.ids.shape[1]


class HGTLayer(nn.Module):
    """Parameter container with the reference's names/shapes (hgt.py:27-79).  The math runs in
    :meth:`HGT.forward` through the C ABI."""

    def __init__(self, in_dim, out_dim, ntype2idx, etype2idx, n_heads, dropout=0.2, use_norm=True,
                 two_stream=False, attn_drop=0.2):
        super().__init__()
        assert out_dim % n_heads == 0
        self.in_dim, self.out_dim, self.n_heads = in_dim, out_dim, n_heads
        self.d_k = out_dim // n_heads
        self.num_types, self.num_relations = len(ntype2idx), len(etype2idx)
        self.use_norm, self.two_stream = use_norm, two_stream
        mk = lambda i, o: nn.ModuleList([nn.Linear(i, o) for _ in range(self.num_types)])
        self.k_linears, self.q_linears, self.v_linears = mk(in_dim, out_dim), mk(in_dim, out_dim), mk(in_dim, out_dim)
        self.a_linears = mk(out_dim, out_dim)
        self.norms = nn.ModuleList([nn.LayerNorm(out_dim) for _ in range(self.num_types)] if use_norm else [])
        self.relation_pri = nn.Parameter(torch.ones(self.num_relations, n_heads))
        self.relation_att = nn.Parameter(torch.empty(self.num_relations, n_heads, self.d_k, self.d_k))
        self.relation_msg = nn.Parameter(torch.empty(self.num_relations, n_heads, self.d_k, self.d_k))
        self.skip = nn.Parameter(torch.ones(self.num_types))
        nn.init.xavier_uniform_(self.relation_att)
        nn.init.xavier_uniform_(self.relation_msg)


def prepare_hgt_weights(sd: Dict[str, torch.Tensor], n_layers: int, n_heads: int, store: CodeStore,
                        device, prefix: str = "", fold_codec: bool = True):
    """Fold / absorb the reference weights (state-dict names of hgt.py) into the layout of
    ``gnnlm_hgt_layer_t``.  Returns (list of dicts of device tensors, dict of codec tensors)."""
    f64 = lambda t: t.detach().to("cpu", torch.float64)
    dev32 = lambda t: t.to(torch.float32).contiguous().to(device)
    d = sd[prefix + "gcs.0.q_linears.0.weight"].shape[0]
    H, dk = n_heads, d // n_heads
    # fold_codec=False: layer 0 sees explicit ntgt states (input adapters, hgt.py:505-507), nothing of the codec is folded
    A = f64(store.A) if (store.A is not None and fold_codec) else None    # [dpq, d]
    bq = f64(store.b) if (A is not None and store.b is not None and store.b.numel() > 0) else None
    layers = []
    for i in range(n_layers):
        p = f"{prefix}gcs.{i}."
        pri, att, msg = f64(sd[p + "relation_pri"]), f64(sd[p + "relation_att"]), f64(sd[p + "relation_msg"])
        out = {}

        def fold(nm, t, rel, scale):
            """W'[hblock,:] = (R_h^T s_h) W[hblock,:],  b'[hblock] = b[hblock] R_h s_h."""
            W, b = f64(sd[p + f"{nm}_linears.{t}.weight"]), f64(sd[p + f"{nm}_linears.{t}.bias"])
            Wn, bn = torch.empty_like(W), torch.empty_like(b)
            for h in range(H):
                R = rel[h] * scale[h]
                Wn[h * dk:(h + 1) * dk] = R.t() @ W[h * dk:(h + 1) * dk]
                bn[h * dk:(h + 1) * dk] = b[h * dk:(h + 1) * dk] @ R
            return Wn, bn

        ones = torch.ones(H, dtype=torch.float64)
        for t, tag in ((0, "t"), (1, "n")):
            out[f"wq_{tag}"] = f64(sd[p + f"q_linears.{t}.weight"])
            out[f"bq_{tag}"] = f64(sd[p + f"q_linears.{t}.bias"])
            out[f"wk_{tag}"], out[f"bk_{tag}"] = fold("k", t, att[0], pri[0] / math.sqrt(dk))
            out[f"wv_{tag}"], out[f"bv_{tag}"] = fold("v", t, msg[0], ones)
            out[f"wa_{tag}"] = f64(sd[p + f"a_linears.{t}.weight"])
            out[f"ba_{tag}"] = f64(sd[p + f"a_linears.{t}.bias"])
            out[f"ln_g_{tag}"] = f64(sd[p + f"norms.{t}.weight"])
            out[f"ln_b_{tag}"] = f64(sd[p + f"norms.{t}.bias"])
        # star edges ('ntgt','inter','tgt'): source type ntgt (1), relation inter (1)
        Wk, Wv, bv = f64(sd[p + "k_linears.1.weight"]), f64(sd[p + "v_linears.1.weight"]), f64(sd[p + "v_linears.1.bias"])
        pre = A if (i == 0 and A is not None) else None
        din = pre.shape[0] if pre is not None else d
        wku = torch.empty(H, din, dk, dtype=torch.float64)
        wvz_t = torch.empty(H, dk, din, dtype=torch.float64)
        bvz = torch.empty(d, dtype=torch.float64)
        for h in range(H):
            Ks = Wk[h * dk:(h + 1) * dk].t() @ att[1][h] * (pri[1][h] / math.sqrt(dk))      # [d, dk]
            Vs = Wv[h * dk:(h + 1) * dk].t() @ msg[1][h]                                      # [d, dk]
            bvh = bv[h * dk:(h + 1) * dk] @ msg[1][h]
            if pre is not None:
                if bq is not None:
                    bvh = bvh - (bq @ pre) @ Vs
                Ks, Vs = pre @ Ks, pre @ Vs
            wku[h], wvz_t[h], bvz[h * dk:(h + 1) * dk] = Ks, Vs.t(), bvh
        out["wku"], out["wvz_t"], out["bvz"] = wku, wvz_t, bvz
        if pre is not None and n_layers > 1:
            # layer 0's ntgt projections on the DECODED rows: h = (x - b) A, q = h Wq^T + bq = x (Wq A^T)^T + (bq - (b A) Wq^T)
            for nm in "qkv":
                W, bb = out[f"w{nm}_n"], out[f"b{nm}_n"]
                out[f"w{nm}_n0"] = W @ pre.t()
                out[f"b{nm}_n0"] = bb - (bq @ pre) @ W.t() if bq is not None else bb.clone()
        lay = {k: dev32(v) for k, v in out.items()}
        # the three tgt projections read the same input: stored back to back ([3d, d] / [3d]) so that the C side can run
        # them as ONE GEMM with N = 3d (it checks the pointers; separate tensors still work)
        qkv_w = torch.cat([lay["wq_t"], lay["wk_t"], lay["wv_t"]]).contiguous()
        qkv_b = torch.cat([lay["bq_t"], lay["bk_t"], lay["bv_t"]]).contiguous()
        for j, nm in enumerate("qkv"):
            lay[f"w{nm}_t"], lay[f"b{nm}_t"] = qkv_w[j * d:(j + 1) * d], qkv_b[j * d:(j + 1) * d]
        layers.append((lay, din))
    codec = {"centroids": store.centroids.to(device, torch.float32).contiguous()}
    if A is not None:
        codec["opq_at"] = dev32(A.t())
        if bq is not None:
            codec["opq_nba"] = dev32(-(bq @ A))
    return layers, codec


def _group_assign(flat_ids, n_store, slot_of, cache=None):
    """gnnlm_group_assign on the current stream -> (group_ids int64 [n], group_slot int32 [n] | None, group_index int32 [n],
    counters int32 [4] on the device: [0] = number of groups).  No host synchronisation."""
    n, dev = flat_ids.numel(), flat_ids.device
    d = _lib.gnnlm_group_assign_t()
    group_ids = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    group_index = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    counters = torch.empty(4, dtype=torch.int32, device=dev)
    group_slot = None
    d.ids, d.n, d.n_store = flat_ids.data_ptr(), n, n_store
    d.slot_of, d.group_ids, d.group_index, d.counters = slot_of.data_ptr(), group_ids.data_ptr(), group_index.data_ptr(), counters.data_ptr()
    if cache is not None:
        group_slot = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        d.id_of_slot, d.cache_state, d.cache_cap, d.group_slot = cache.id_of_slot.data_ptr(), cache.state.data_ptr(), cache.capacity, group_slot.data_ptr()
    _lib.call_desc("gnnlm_group_assign", d)
    return group_ids, group_slot, group_index, counters


class CentreStateCache:
    """Cross-batch cache of the context groups' centre states (``gnnlm_hgt_io_t.state_cache``).

    The star edges of layer l >= 1 read the centre node of a context group after l ntgt updates; ntgt nodes never receive from
    tgt nodes (token_block_dataset.py:395-398: only `ntgt -> tgt` and `ntgt <-> ntgt` edges), so that state is a function of
    the centre's datastore row alone -- the same for every token, block and batch that retrieves the row.  Real neighbour lists
    repeat rows heavily (the reference's own "todo: merge same nodes", :355, is the within-block half of this).  The cache
    keeps ``[n_layers - 1, capacity, d]`` float32 states in HBM and a direct row -> slot table (int32 per datastore row: no
    hashing).  Slots are filled in arrival order in TWO GENERATIONS (the halves of the table): when the current half is full
    the other one is emptied and refilled, so the more recent half of what was computed always survives.  Exact: the states
    are the ones the un-cached call computes.

    Everything that decides -- hit or miss, the slot of a new row, the generation switch -- runs on the DEVICE
    (``gnnlm_group_assign``, csrc/groups.hip): the bookkeeping (`state`: half being filled, entries per half, switches, rows
    computed) lives in device memory and a call never synchronises; `stats` / `used` read it back (and do)."""

    def __init__(self, n_store, n_layers, d, capacity, device, code_bytes=0):
        self.n_store, self.capacity, self.device = n_store, int(capacity), device
        self.half = self.capacity // 2
        self.slot_of = torch.full((n_store,), -1, dtype=torch.int32, device=device)
        self.id_of_slot = torch.empty(self.capacity, dtype=torch.int64, device=device)
        self.states = torch.empty(n_layers - 1, self.capacity, d, dtype=torch.float32, device=device)
        # sharded store: the PQ code row of every cached centre (layer 0's star edges read the code of every neighbour, and only the
        # rows the cache lacks are fetched from their owners)
        self.codes = torch.zeros(self.capacity, code_bytes, dtype=torch.uint8, device=device) if code_bytes else None
        self.state = torch.zeros(8, dtype=torch.int32, device=device)       # gnnlm_group_assign_t.cache_state
        self.stream = _lib.raw_stream(device)                                # slots are reused in stream order: one stream only
        self._host = {"lookups": 0, "groups": 0, "clears": 0, "computed0": 0}

    @property
    def stats(self):
        st = self.state.tolist()                                             # (synchronises: a reporting path)
        return {"lookups": self._host["lookups"], "groups": self._host["groups"], "computed": st[4] - self._host["computed0"],
                "generations": 1 + st[3] + self._host["clears"]}

    def reset_stats(self):
        self._host.update(lookups=0, groups=0, computed0=int(self.state[4].item()))

    @property
    def used(self):
        st = self.state.tolist()
        return st[1] + st[2]

    def clear(self):
        self.slot_of.fill_(-1)
        self.state[:3] = 0
        self._host["clears"] += 1

    def assign(self, flat_ids):
        """flat_ids int64 [n]: neighbour rows (-1 / out of range: none) -> (rows to compute int64 [n] of which the first
        counters[0] count, their slots int32 [n], slot of every neighbour int32 [n] (-1: not a neighbour), counters), or None
        when a batch of n neighbours may not fit one generation (n > capacity / 2: the caller merges within the batch only)."""
        if flat_ids.numel() > self.half:
            return None
        out = _group_assign(flat_ids, self.n_store, self.slot_of, self)
        self._host["lookups"] += 1
        self._host["groups"] += int(flat_ids.numel())
        return out


class HGT(nn.Module):
    """Drop-in for ``HGT`` of fairseq/models/hgt.py:459-513 (eval forward, non-incremental)."""

    def __init__(self, ntype2idx=None, etype2idx=None, in_dim=1024, hidden_dim=1024, out_dim=1024, n_layers=3,
                 n_heads=8, use_norm=True, dropout=0.0, two_stream=False, attn_drop=0.0):
        super().__init__()
        ntype2idx = ntype2idx or NTYPE2IDX
        etype2idx = etype2idx or ETYPE2IDX
        if dict(ntype2idx) != NTYPE2IDX or dict(etype2idx) != ETYPE2IDX:
            raise NotImplementedError("only the node/edge types of TokenGraphTransformerDecoder "
                                      "(transformer.py:913-920) are supported")
        if two_stream or not use_norm:
            raise NotImplementedError("two_stream=True / use_norm=False are not on the eval path (transformer.py:931)")
        self.ntype2idx, self.etype2idx = ntype2idx, etype2idx
        self.in_dim, self.hidden_dim, self.out_dim = in_dim, hidden_dim, out_dim
        self.n_layers, self.n_heads = n_layers, n_heads
        self.gcs = nn.ModuleList([HGTLayer(hidden_dim, hidden_dim, ntype2idx, etype2idx, n_heads,
                                           use_norm=use_norm, dropout=dropout, attn_drop=attn_drop)
                                  for _ in range(n_layers)])
        # hgt.py:476-479,491-492: input adapters (one Linear per node type, followed by GELU) when in_dim != hidden_dim,
        # an output Linear when hidden_dim != out_dim
        self.adapt_ws = nn.ModuleList([nn.Linear(in_dim, hidden_dim) for _ in ntype2idx] if in_dim != hidden_dim else [])
        if hidden_dim != out_dim:
            self.out = nn.Linear(hidden_dim, out_dim)
        self._prepared = self._plist = None
        self.gemm_precision = 0     # 0 exact f32 MFMA | 1 bf16x3 | 2 bf16x6 (opt-in split-bf16 emulation)
        self.dedup_groups = os.environ.get("GNNLM_DEDUP", "1") != "0"      # merge equal context groups of a batch (multi-layer models)
        self.dedup_rows = os.environ.get("GNNLM_DEDUP_ROWS", "1") != "0"   # ... and project layer 0's K / V once per distinct datastore ROW (ABI 11)
        self._last_groups = None                                             # (groups of the last batch, device counter of the computed ones)
        self._merge_tables = {}                                              # (device, n_store, stream) -> row -> group table of the within-batch merge
        # centre states of context groups kept ACROSS batches (CentreStateCache): HBM budget in GiB, 0 = off
        self.state_cache_gib = float(os.environ.get("GNNLM_STATE_CACHE_GIB", "32"))
        self.state_cache_slots = None
        self.state_cache = None

    @property
    def last_groups(self):
        """(context groups of the last batch, groups its ntgt pipeline ran over); reads a device counter (synchronises)."""
        if self._last_groups is None:
            return None
        n, counters = self._last_groups
        return n, (int(counters[0].item()) if torch.is_tensor(counters) else counters)

    def _load_from_state_dict(self, *a, **k):
        self._prepared = self._plist = None
        return super()._load_from_state_dict(*a, **k)

    def prepare(self, store: CodeStore, device):
        """Fold the weights for ``store``'s codec and build the C descriptors (cached)."""
        # The folded weights depend on the parameters and on the store's codec (A / b are folded into layer 0):
        # key on their storage + in-place version counters, and keep the store alive inside the cache entry so
        # that neither an id() reuse nor an in-place edit can serve stale weights.  Table pointers (codes / vals)
        # are NOT part of the key: forward() refreshes them on every call.
        ver = lambda t: None if t is None else (t.data_ptr(), t._version, tuple(t.shape))
        if self._plist is None:                                   # (walking the module tree costs ~50 us per call: kept, dropped by load_state_dict)
            self._plist = list(self.parameters())
        key = (str(device), ver(store.centroids), ver(store.A), ver(store.b),
               tuple(ver(p) for p in self._plist))
        if self._prepared is not None and self._prepared["key"] == key:
            self._bind_store(self._prepared["model"], store)
            self._prepared["store"] = store
            return self._prepared
        layers, codec = prepare_hgt_weights(self.state_dict(), self.n_layers, self.n_heads, store, device,
                                            fold_codec=(self.in_dim == self.hidden_dim))
        arr = (_lib.gnnlm_hgt_layer_t * self.n_layers)()
        for i, (w, din) in enumerate(layers):
            for name, t in w.items():
                setattr(arr[i], name, t.data_ptr())
            arr[i].din = din
        m = _lib.gnnlm_hgt_t()
        m.d, m.n_heads, m.n_layers = self.hidden_dim, self.n_heads, self.n_layers
        m.ln_eps = self.gcs[0].norms[0].eps
        M, _, dsub = store.centroids.shape
        m.M, m.dsub = M, dsub
        m.centroids = codec["centroids"].data_ptr()
        if "opq_at" in codec:
            m.opq_at = codec["opq_at"].data_ptr()
        if "opq_nba" in codec:
            m.opq_nba = codec["opq_nba"].data_ptr()
        self._bind_store(m, store)
        m.layers = ctypes.cast(arr, ctypes.c_void_p)
        self._prepared = {"key": key, "model": m, "layers_arr": arr, "tensors": (layers, codec), "ws": None,
                          "store": store}
        return self._prepared

    def _state_cache_for(self, prep, G, device):
        """The centre-state cache of (these weights, this store, this graph shape), or None (off, another stream)."""
        if self.state_cache_gib <= 0 or self.n_layers < 2:
            return None
        store = G.store
        # every cached group is taken for a VALID neighbour (the cached path carries no per-group validity): cache only when every
        # in-range row can be read -- the whole table resident, its shards mapped, or a fetcher that brings what is not local.  A
        # partial local table without any of these runs un-cached, where rows that are not local are excluded from the star softmax.
        if G.fetcher is None and shards_device_ptr(store) is None and not (store.row0 == 0 and store.codes.shape[0] >= store.n_store):
            return None
        c = self.state_cache
        # everything the cached states are a function of: the folded weights (prep key: parameters + codec), the GEMM arithmetic,
        # the context shape, and the code table itself (identity AND in-place version)
        key = (prep["key"], id(store), store.codes.data_ptr(), store.codes._version, store.n_store, self.gemm_precision,
               G.left, G.right, G.max_intra_context, G.fetcher is not None)
        if c is not None and (getattr(c, "key", None) != key):
            c = self.state_cache = None                                       # other weights / store / arithmetic: the states are stale
        if c is None:
            if torch.cuda.is_current_stream_capturing():
                return None                                                   # (persistent tables are not born inside a capture)
            code_bytes = store.codes.shape[1] if G.fetcher is not None else 0
            per = (self.n_layers - 1) * self.hidden_dim * 4 + 8 + code_bytes
            budget = self.state_cache_gib * 2 ** 30
            if self.state_cache_slots is None:                                # never more than half of what is free right now
                budget = min(budget, 0.5 * torch.cuda.mem_get_info(device)[0])
            cap = int(min(store.n_store, max(0, budget - 4 * store.n_store) // per))
            if self.state_cache_slots is not None:                            # explicit capacity (tests, tuning)
                cap = int(min(2 * store.n_store, self.state_cache_slots))
            if cap < 2:
                return None
            try:
                c = CentreStateCache(store.n_store, self.n_layers, self.hidden_dim, cap, device, code_bytes)
            except torch.OutOfMemoryError:
                self.state_cache_gib = 0.0                                    # run un-cached rather than fail
                return None
            self.state_cache = c
            c.key = key
        if c.stream != _lib.raw_stream(device):
            return None
        return c

    def _merge_table(self, n_store, device):
        """Row -> group table of the within-batch merge (int32 per datastore row, all -1 between calls), one per stream.
        Persistent tables (this one, the centre-state cache) are never born inside a HIP-graph capture -- they would live in the
        graph's private pool: a step captured on a stream that has not run an eager forward yet is captured UN-merged (same
        results).  To capture the merged step, run one eager forward on a stream and capture on it
        (`torch.cuda.graph(g, stream=that_stream)`)."""
        key = (str(device), n_store, _lib.raw_stream(device))
        t = self._merge_tables.get(key)
        if t is None:
            if torch.cuda.is_current_stream_capturing():
                return None
            t = self._merge_tables[key] = torch.full((n_store,), -1, dtype=torch.int32, device=device)
        return t

    def _row_table(self, n_store, device):
        """Second row table (int32 per datastore row, all -1 between calls) of the ROW-keyed layer-0 K / V projections (ABI 11):
        like the merge table, one per stream and never born inside a capture."""
        key = (str(device), n_store, _lib.raw_stream(device), "rows")
        t = self._merge_tables.get(key)
        if t is None:
            if torch.cuda.is_current_stream_capturing():
                return None
            t = self._merge_tables[key] = torch.full((n_store,), -1, dtype=torch.int32, device=device)
        return t

    @staticmethod
    def _bind_store(m, store):
        """(Re)point the descriptor at the store's tables: cheap, done on every forward."""
        m.codes = store.codes.data_ptr()
        m.vals, m.vals_itemsize = (store.vals.data_ptr(), store.vals.element_size()) if store.vals is not None else (None, 4)
        m.n_store, m.row0, m.n_local = store.n_store, store.row0, store.codes.shape[0]
        m.shards = shards_device_ptr(store)

    def invalidate(self):
        """Drop the prepared (folded) weights, e.g. after swapping parameter tensors by hand."""
        self._prepared = self._plist = None

    def release_stream_state(self, keep=()):
        """Free what is held PER STREAM (the row -> group merge tables, 4 bytes per datastore row each, and the forward's scratch
        arenas) for every stream but `keep` (raw handles): a driver that scores on its own side streams calls this when its run
        ends -- torch hands out 32 pooled stream handles, a 103 M-row store would otherwise pin 0.4 GB per handle ever used."""
        keep = set(keep)
        for k_ in [k_ for k_ in self._merge_tables if k_[2] not in keep]:
            del self._merge_tables[k_]
        if self._prepared is not None and self._prepared.get("ws"):
            for k_ in [k_ for k_ in self._prepared["ws"] if k_ not in keep]:
                del self._prepared["ws"][k_]

    def forward(self, G: NeighborGraph, features: Dict[str, torch.Tensor] = None, etypes=None,
                incremental_state=None, return_ntgt: bool = False):
        """Returns ``{'tgt': [n_blocks*T, d]}`` (+ ``'ntgt'`` [n_valid_slots, d] in reference node
        order when ``return_ntgt``; the eval path never consumes it, transformer.py:1053)."""
        if incremental_state is not None:
            raise NotImplementedError("incremental decoding (HGTLayer.infer) is out of scope (generation only)")
        tgt = (features or {}).get("tgt", G.tgt_h)
        if tgt is None:
            raise ValueError("tgt features missing: pass features={'tgt': ...} or G.tgt_h")
        if not tgt.is_cuda:
            raise _lib.GnnlmError("HGT.forward needs device tensors; gnnlm_amd has no CPU fallback")
        tgt = tgt.to(torch.float32).contiguous()
        adapted = self.in_dim != self.hidden_dim
        n_g = 1 + G.left + G.right
        if adapted:
            # F.gelu(adapt_ws[ntype](feat)) (hgt.py:505-507): the ntgt features are the decoded PQ rows of every slot
            # (transformer.py:1043-1045) -- decoded here explicitly, a non-linearity sits between them and the layer
            from . import ops
            if G.fetched_codes is not None:
                raise NotImplementedError("input adapters with a sharded store (fetched codes) are not built")
            st = G.store
            lin = lambda x, mod: ops.gemm_nt(x, mod.weight.detach().to(tgt.device).contiguous(),
                                             bias=mod.bias.detach().to(tgt.device).contiguous())
            dec = ops.pq_gather_decode(st.codes, st.centroids, G.ids.reshape(-1).contiguous(), G.left, G.right,
                                       n_store=st.n_store, row0=st.row0)
            x0 = dec["x"]
            if st.A is not None:                                        # (x - b) @ A  (pq_wrapper.py:198-202)
                At = st.A.t().contiguous()
                x0 = ops.gemm_nt(x0, At, bias=None if st.b is None or st.b.numel() == 0 else -(st.b @ st.A).contiguous())
            ntgt0 = ops.gelu_(lin(x0, self.adapt_ws[self.ntype2idx["ntgt"]]))
            ntgt0 *= dec["valid"].to(torch.float32)[:, None]            # rows of invalid slots stay zero (they are not nodes)
            ntgt_valid = dec["valid"]
            tgt = ops.gelu_(lin(tgt, self.adapt_ws[self.ntype2idx["tgt"]]))
        prep = self.prepare(G.store, tgt.device)
        m = prep["model"]
        m.left, m.right, m.max_intra_context = G.left, G.right, G.max_intra_context
        m.gemm_precision = self.gemm_precision
        io = _lib.gnnlm_hgt_io_t()
        io.n_blocks, io.T, io.kg = G.n_blocks, G.T, G.kg
        ids = G.ids.contiguous()
        io.tgt_feats, io.ids = tgt.data_ptr(), ids.data_ptr()
        fetched = (G.fetched_codes, G.fetched_valid, G.fetched_index, G.fetched_centres_only)
        st, fetcher = G.store, G.fetcher
        if fetcher is not None and adapted:
            raise NotImplementedError("input adapters with a sharded store are not built")
        if adapted:
            io.ntgt_feats, io.ld_ntgt, io.ntgt_valid = ntgt0.data_ptr(), ntgt0.stride(0), ntgt_valid.data_ptr()
        # exact de-duplication of context groups (the reference's "todo: merge same nodes", token_block_dataset.py:355): the ntgt
        # states of a group depend on its centre row only, and the neighbour lists of nearby tokens overlap heavily -- the
        # multi-layer ntgt pipeline runs once per DISTINCT centre row of the batch, and across batches only for the rows the
        # cache lacks.  Decided on the device (gnnlm_group_assign): no sort, no host round trip, capturable.
        keep = []                                                             # tensors the launch reads: alive until it is enqueued
        cache = None
        self._last_groups = None
        merged = (self.n_layers > 1 and self.dedup_groups and not adapted and not return_ntgt and G.fetched_codes is None)
        if merged:
            flat = ids.reshape(-1)
            cache = self._state_cache_for(prep, G, tgt.device)
            hit = cache.assign(flat) if cache is not None else None
            if hit is not None:
                # across batches: only the groups the cache lacks are computed; every neighbour reads its group's slot
                group_ids, group_slot, group_index, counters = hit
                io.group_slot = group_slot.data_ptr()
                io.state_cache, io.cache_cap = cache.states.data_ptr(), cache.capacity
                if cache.codes is not None:
                    io.code_cache = cache.codes.data_ptr()
            else:
                cache = None
                table = self._merge_table(st.n_store, tgt.device)
                if table is None:
                    merged = False
                else:
                    group_ids, _, group_index, counters = _group_assign(flat, st.n_store, table)
        L = _lib.lib()
        try:                                  # (from here on a failure leaves assigned cache slots without states: the cache is cleared)
            if merged:
                io.group_ids, io.n_unique, io.group_index = group_ids.data_ptr(), flat.numel(), group_index.data_ptr()
                io.n_unique_dev = counters.data_ptr()
                if self.dedup_rows and fetcher is None and (G.left or G.right):
                    # layer 0's K / V once per distinct datastore ROW among the groups' slots (neighbouring groups share rows)
                    rt = self._row_table(st.n_store, tgt.device)
                    if rt is not None:
                        io.row_table = rt.data_ptr()
                        keep.append(rt)
                keep += [group_ids, group_index, counters, hit]
                self._last_groups = (flat.numel(), counters)
                if fetcher is not None:                                           # every distinct centre row is requested ONCE
                    fetched = fetcher.fetch_groups(group_ids, G.left, G.right, counters) + (False,)
            elif fetcher is not None and G.fetched_codes is None:
                centres_only = self.n_layers == 1 and not return_ntgt
                fetched = fetcher.fetch_codes(ids, G.left, G.right, centres_only) + (centres_only,)
            if fetched[0] is not None:
                # (an EMPTY answer -- every group of the batch was a cache hit -- has no address, and a null fetched_codes means "read the
                # local table" to the C side: name real memory)
                real = lambda t, dt: t if t.numel() else torch.zeros(16, dtype=dt, device=tgt.device)
                fc, fv, fi = real(fetched[0], torch.uint8), fetched[1], fetched[2]
                io.fetched_codes = fc.data_ptr()
                if fv is not None:
                    fv = real(fv, torch.uint8)
                    io.fetched_valid = fv.data_ptr()
                io.fetched_centres_only = int(fetched[3])
                if fi is not None:
                    fi = real(fi, torch.int32)
                    io.fetched_index = fi.data_ptr()
                keep.append((fc, fv, fi))
            out_tgt = torch.empty_like(tgt)
            io.out_tgt = out_tgt.data_ptr()
            S = ids.shape[0] * G.kg * n_g
            if return_ntgt:
                out_ntgt = torch.empty(S, self.hidden_dim, device=tgt.device, dtype=torch.float32)
                out_valid = torch.empty(S, device=tgt.device, dtype=torch.uint8)
                io.out_ntgt, io.out_valid = out_ntgt.data_ptr(), out_valid.data_ptr()
            need = L.gnnlm_hgt_workspace_bytes(ctypes.byref(m), ctypes.byref(io))
            # one arena per stream: concurrent forwards on different streams must not share scratch
            key = _lib.raw_stream()
            if prep["ws"] is None:
                prep["ws"] = {}
            ws = prep["ws"].get(key)
            if ws is None or ws.numel() < need or ws.device != tgt.device:
                prep["ws"][key] = ws = None
                ws = prep["ws"][key] = torch.empty(need, device=tgt.device, dtype=torch.uint8)
            _lib.check(L.gnnlm_hgt_forward(ctypes.byref(m), ctypes.byref(io), _lib.ptr(ws), ws.numel(),
                                           _lib.stream()), "gnnlm_hgt_forward")
        except Exception:
            if cache is not None:             # the new rows own slots whose states were never written: nothing cached survives
                cache.clear()
            raise
        out = {"tgt": out_tgt}
        if return_ntgt:
            out["ntgt"] = out_ntgt[out_valid.bool()]
        if self.hidden_dim != self.out_dim:                               # hgt.py:513
            from . import ops
            W, bo = self.out.weight.detach().to(tgt.device).contiguous(), self.out.bias.detach().to(tgt.device).contiguous()
            out = {k_: ops.gemm_nt(v.contiguous(), W, bias=bo) for k_, v in out.items()}
        return out
