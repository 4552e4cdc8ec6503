"""Oracle: one eval block end to end on the CPU (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Follows the reference's hot loop as written:
  GraphTokenBlockDataset.__getitem__  (token_block_dataset.py:287-331)   gathers + graph
  TokenGraphTransformerDecoder.forward / extract_graph_features (transformer.py:943-1053)
  AdaptiveSoftmax.get_log_prob + gather (adaptive_softmax.py:170-206, sequence_scorer.py:89)
  KNNModel.get_knn_prob (knn_model.py:103-217) with the search results supplied
  combine_knn_and_vocab_probs (sequence_scorer.py:55-68)
"""
import numpy as np
import torch

from . import adaptive_softmax as oas
from . import graph as og
from . import hgt as ohgt
from . import knn as oknn
from . import pq as opq


def gather_block(neighbor_idxs, codes, vals, n_store, left, right, reference_loop=False, tgt_offsets=None,
                 invalid_neighbor_context=0):
    """Row gathers of one block: node codes / labels in reference node order + the graph.

    ``reference_loop=True`` uses the per-row Python loop of token_block_dataset.py:354-400;
    otherwise the vectorised slot layout (identical result, test_oracle_graph checks it)."""
    T = neighbor_idxs.shape[0]
    pos = np.zeros(T, np.int64) if tgt_offsets is None else np.asarray(tgt_offsets, dtype=np.int64)
    c = invalid_neighbor_context if tgt_offsets is not None else 0         # :361 (train split only, language_modeling.py:299)
    if reference_loop:
        g = og.build_graph(neighbor_idxs, pos, n_store, left, right, invalid_neighbor_context=c)
        rows = g["ntgt_offsets"]
        return g, codes[rows], vals[rows]
    g = og.build_graph(neighbor_idxs, pos, n_store, left, right, invalid_neighbor_context=c)
    rows, valid = og.slot_layout(neighbor_idxs, n_store, left, right, pos, c)
    flat = rows[valid]
    return g, codes[flat], vals[flat]          # codes / vals: anything indexable by global rows (array, memmap, HostRows)


def hgt_block(sd, n_layers, n_heads, tgt_feats, ntgt_codes, graph, centroids, A, b,
              dtype=torch.float32, return_all_layers=False):
    """PQ decode of the ntgt nodes (transformer.py:1043-1045) + HGT (transformer.py:1050)."""
    ntgt = opq.pq_decode(ntgt_codes, centroids, A, b)
    feats = {"tgt": torch.as_tensor(np.asarray(tgt_feats, dtype=np.float32)).to(dtype),
             "ntgt": torch.as_tensor(ntgt).to(dtype)}
    sdd = {k: v.to(dtype) for k, v in sd.items()}
    return ohgt.hgt_forward(sdd, n_layers, n_heads, feats, graph, return_all_layers)


def eval_block(blk, model, lmbda, temperature, dtype=torch.float32):
    """Score one block.  ``blk``: dict(neighbor_idxs [T,kg], tgt_feats [T,d] fp16/fp32,
    targets [T], knn_sims [T,k], knn_ids [T,k][, tgt_offsets [T]: the tokens' global offsets, train split]);
    ``model``: dict(sd, n_layers, n_heads, centroids, A, b, codes, vals, n_store, left, right, asm
    [, invalid_neighbor_context]).
    Returns per-token dict(gcn_feat, lm_logp, p_knn, recall, logp)."""
    g, ncodes, _ = gather_block(blk["neighbor_idxs"], model["codes"], model["vals"],
                                model["n_store"], model["left"], model["right"], tgt_offsets=blk.get("tgt_offsets"),
                                invalid_neighbor_context=model.get("invalid_neighbor_context", 0))
    h = hgt_block(model["sd"], model["n_layers"], model["n_heads"], blk["tgt_feats"], ncodes, g,
                  model["centroids"], model["A"], model["b"], dtype)
    x = h["tgt"]
    tgt = torch.as_tensor(blk["targets"]).long()
    asm = {k: ([None if e is None else e.to(dtype) for e in v] if isinstance(v, list) and k != "cutoff"
               else (v.to(dtype) if torch.is_tensor(v) else v)) for k, v in model["asm"].items()}
    lm_logp = oas.target_log_prob(x, tgt, asm).float()
    out = {"gcn_feat": x, "lm_logp": lm_logp}
    if lmbda > 0:
        p_knn, recall = oknn.knn_target_prob(blk["knn_sims"], blk["knn_ids"], model["vals"],
                                             tgt, temperature)
        out["p_knn"], out["recall"] = p_knn, recall
        out["logp"] = oknn.combine_knn_and_vocab_probs(p_knn, lm_logp, lmbda)
    else:
        out["logp"] = lm_logp
    return out


def run_problem(prob, lmbda, temperature, dtype=torch.float32):
    """Evaluate a ``gnnlm_amd.synthetic.make_problem`` problem with the oracle, block by block."""
    blk = prob["block"]
    model = {"sd": {k: v for k, v in prob["sd"].items()}, "n_layers": prob["n_layers"], "n_heads": prob["n_heads"],
             "centroids": prob["cen"], "A": prob["A"], "b": prob["b"], "codes": prob["codes"], "vals": prob["vals"],
             "n_store": prob["n_store"], "left": prob["left"], "right": prob["right"], "asm": prob["asm"]}
    T, outs = blk["T"], []
    for i in range(blk["n_blocks"]):
        sl = slice(i * T, (i + 1) * T)
        one = {"neighbor_idxs": blk["ids"][sl], "tgt_feats": blk["tgt_feats"][sl], "targets": blk["targets"][sl],
               "knn_sims": blk["knn_sims"][sl], "knn_ids": blk["knn_ids"][sl]}
        outs.append(eval_block(one, model, lmbda, temperature, dtype))
    return {k: torch.cat([o[k] for o in outs]).numpy() for k in outs[0]}
