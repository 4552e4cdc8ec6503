"""Oracle: Heterogeneous Graph Transformer forward (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Restates ``HGTLayer.forward`` (fairseq/models/hgt.py:299-420, non-two-stream branch) and
``HGT.forward`` (:494-513) in torch-CPU over explicit edge lists, with the DGL primitives it
uses spelled out:

  * ``fn.v_dot_u('q','k','t')`` then ``.sum(-1)``      -> per-edge, per-head dot  (:354-355)
  * ``edge_softmax(g, score, norm_by='dst')``          -> softmax over the incoming edges of
                                                          each dst node, per head  (:356)
  * ``multi_update_all({etype: (u_mul_e, sum)}, cross_reducer='mean')``
        -> per edge type: sum of alpha * v' over incoming edges (0 for zero in-degree), then the
           MEAN over the edge types that target the node type                          (:383-386)

Weights use the reference's state-dict names (``gcs.{i}.k_linears.{t}.weight`` ...), node type
ids ``tgt=0, ntgt=1`` and relation ids ``intra=0, inter=1`` (transformer.py:913-920).
"""
import math
from typing import Dict, List, Tuple

import torch

NTYPE2IDX = {"tgt": 0, "ntgt": 1}
ETYPE2IDX = {"intra": 0, "inter": 1}
HGT_ETYPES = [("tgt", "intra", "tgt"), ("ntgt", "inter", "tgt"), ("ntgt", "intra", "ntgt")]


def init_hgt_weights(n_layers, d, n_heads, seed=0, dtype=torch.float32, scale_bias=0.02,
                     random_pri=True):
    """Random HGT state dict with the reference's key names / shapes (hgt.py:55-79).

    Linear weights: uniform(+-1/sqrt(d)) like nn.Linear's default; relation_att/msg: Xavier
    uniform (hgt.py:78-79); LayerNorm gamma/beta and relation_pri perturbed away from their
    init values so a test cannot pass with them ignored."""
    g = torch.Generator().manual_seed(seed)
    d_k = d // n_heads
    sd = {}

    def u(*shape, bound):
        return ((torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1) * bound).to(dtype)

    for i in range(n_layers):
        for t in range(2):
            for nm in ("k", "q", "v", "a"):
                sd[f"gcs.{i}.{nm}_linears.{t}.weight"] = u(d, d, bound=1 / math.sqrt(d))
                sd[f"gcs.{i}.{nm}_linears.{t}.bias"] = u(d, bound=scale_bias)
            sd[f"gcs.{i}.norms.{t}.weight"] = (1 + u(d, bound=0.1))
            sd[f"gcs.{i}.norms.{t}.bias"] = u(d, bound=0.1)
        xb = math.sqrt(6.0 / (n_heads * d_k * d_k + d_k * d_k))  # xavier on [R,H,dk,dk] fan calc
        sd[f"gcs.{i}.relation_att"] = u(2, n_heads, d_k, d_k, bound=max(xb, 1 / math.sqrt(d_k)))
        sd[f"gcs.{i}.relation_msg"] = u(2, n_heads, d_k, d_k, bound=max(xb, 1 / math.sqrt(d_k)))
        pri = torch.ones(2, n_heads, dtype=dtype)
        if random_pri:
            pri = pri + u(2, n_heads, bound=0.3)
        sd[f"gcs.{i}.relation_pri"] = pri
        sd[f"gcs.{i}.skip"] = torch.ones(2, dtype=dtype)
    return sd


def _edge_softmax_by_dst(score, dst, num_dst):
    """score [E,H] -> softmax over edges sharing a dst (per head)."""
    H = score.shape[1]
    mx = torch.full((num_dst, H), -float("inf"), dtype=score.dtype)
    mx = mx.scatter_reduce(0, dst[:, None].expand(-1, H), score, reduce="amax", include_self=True)
    e = torch.exp(score - mx[dst])
    den = torch.zeros((num_dst, H), dtype=score.dtype).index_add_(0, dst, e)
    return e / den[dst]


def hgt_layer_forward(sd: Dict[str, torch.Tensor], layer: int, n_heads: int,
                      h: Dict[str, torch.Tensor],
                      edges: Dict[Tuple[str, str, str], Tuple[torch.Tensor, torch.Tensor]],
                      eps: float = 1e-5) -> Dict[str, torch.Tensor]:
    """One HGTLayer.forward (hgt.py:299-420)."""
    p = f"gcs.{layer}."
    d = next(iter(h.values())).shape[1]
    d_k = d // n_heads
    sqrt_dk = math.sqrt(d_k)
    q, k, v = {}, {}, {}
    for nt, x in h.items():                                            # :315-322
        t = NTYPE2IDX[nt]
        lin = lambda nm: torch.nn.functional.linear(x, sd[p + f"{nm}_linears.{t}.weight"],
                                                    sd[p + f"{nm}_linears.{t}.bias"])
        k[nt] = lin("k").view(-1, n_heads, d_k)
        v[nt] = lin("v").view(-1, n_heads, d_k)
        q[nt] = lin("q").view(-1, n_heads, d_k)

    per_dst: Dict[str, List[torch.Tensor]] = {nt: [] for nt in h}
    for (st, et, dt_), (src, dst) in edges.items():
        e_id = ETYPE2IDX[et]
        rel_att = sd[p + "relation_att"][e_id]
        rel_pri = sd[p + "relation_pri"][e_id]
        rel_msg = sd[p + "relation_msg"][e_id]
        kk = torch.einsum("bij,ijk->bik", k[st], rel_att)              # :347
        vv = torch.einsum("bij,ijk->bik", v[st], rel_msg)              # :348
        n_dst = h[dt_].shape[0]
        if src.numel() == 0:
            per_dst[dt_].append(torch.zeros(n_dst, n_heads, d_k, dtype=kk.dtype))
            continue
        score = (q[dt_][dst] * kk[src]).sum(-1) * rel_pri / sqrt_dk     # :354-355
        alpha = _edge_softmax_by_dst(score, dst, n_dst)                # :356 (dropout = id in eval)
        m = torch.zeros(n_dst, n_heads, d_k, dtype=kk.dtype)
        m.index_add_(0, dst, vv[src] * alpha.unsqueeze(-1))            # :383-385
        per_dst[dt_].append(m)

    new_h = {}
    for nt, x in h.items():                                            # :397-407
        t = NTYPE2IDX[nt]
        agg = torch.stack(per_dst[nt], 0).mean(0)                      # cross_reducer='mean' :386
        out = torch.nn.functional.linear(agg.reshape(-1, d), sd[p + f"a_linears.{t}.weight"],
                                         sd[p + f"a_linears.{t}.bias"]) + x
        new_h[nt] = torch.nn.functional.layer_norm(out, (d,), sd[p + f"norms.{t}.weight"],
                                                   sd[p + f"norms.{t}.bias"], eps)
    return new_h


def hgt_forward(sd, n_layers, n_heads, feats, graph, return_all_layers=False):
    """HGT.forward (hgt.py:494-513), incl. the input adapters `F.gelu(adapt_ws[ntype](feat))` (:505-507) and the
    output Linear (:513) when the state dict carries them (in_dim != hidden_dim / hidden_dim != out_dim).

    ``graph`` is the dict returned by :func:`oracle.graph.build_graph`."""
    ten = lambda a: torch.as_tensor(a, dtype=torch.int64)
    edges = {
        ("tgt", "intra", "tgt"): tuple(map(ten, graph["intra_tgt"])),
        ("ntgt", "inter", "tgt"): tuple(map(ten, graph["inter"])),
        ("ntgt", "intra", "ntgt"): tuple(map(ten, graph["intra_ntgt"])),
    }
    h = dict(feats)
    if "adapt_ws.0.weight" in sd:
        h = {nt: torch.nn.functional.gelu(torch.nn.functional.linear(x, sd[f"adapt_ws.{NTYPE2IDX[nt]}.weight"],
                                                                      sd[f"adapt_ws.{NTYPE2IDX[nt]}.bias"]))
             for nt, x in h.items()}
    outs = []
    for i in range(n_layers):
        h = hgt_layer_forward(sd, i, n_heads, h, edges)
        outs.append(h)
    if "out.weight" in sd:
        outs[-1] = h = {nt: torch.nn.functional.linear(x, sd["out.weight"], sd["out.bias"]) for nt, x in h.items()}
    return outs if return_all_layers else h
