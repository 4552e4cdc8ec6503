"""Oracle: kNN-LM probability and interpolation (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Restates, in torch-CPU float32 (the reference's dtype):
  * sims_from_search -- the sim_func dispatch of KNNModel.get_knn_prob (knn/knn_model.py:137-177)
  * knn_target_prob  -- mask / softmax / target match / recall       (knn/knn_model.py:179-217)
  * brute_force_search -- stand-in for faiss ``index.search`` (knn_model.py:100) on small stores:
                          exact inner product / L2 over the full keys.  faiss ANN results and ADC
                          distances are NOT reproduced (parity unpinned, see oracle/__init__.py).
  * combine_knn_and_vocab_probs -- fairseq/sequence_scorer.py:55-68 with the epsilon of :110,:121
  * perplexity -- fairseq_cli/eval_lm.py:273-274,325-331
"""
import math

import numpy as np
import torch

MASK_VALUE = -1e10          # knn_model.py:193
KNN_EPSILON = 1e-10         # sequence_scorer.py:110


def normalize_queries(queries, cosine):
    """knn_model.py:181-184 -- L2-normalise when the index file name contains 'cosine'."""
    if cosine:
        return queries / (queries ** 2).sum(-1, keepdims=True).sqrt()
    return queries


def brute_force_search(queries, keys, k, metric="ip", cosine=False):
    """Exact top-k (descending similarity).  Returns (dists [n,k] f32, ids [n,k] i64) with the
    faiss sign convention: inner product for 'ip', squared L2 distance for 'l2'."""
    q = np.asarray(queries, dtype=np.float32)
    ks = np.asarray(keys, dtype=np.float32)
    if cosine:
        ks = ks / np.sqrt((ks ** 2).sum(-1, keepdims=True))
    if metric == "ip":
        s = q @ ks.T
        ids = np.argsort(-s, axis=1, kind="stable")[:, :k]
        d = np.take_along_axis(s, ids, 1)
    else:
        s = ((q[:, None, :] - ks[None, :, :]) ** 2).sum(-1)
        ids = np.argsort(s, axis=1, kind="stable")[:, :k]
        d = np.take_along_axis(s, ids, 1)
    return d.astype(np.float32), ids.astype(np.int64)


def sims_from_search(dists, knns, queries, metric_type, keys=None, cosine=False):
    """knn_model.py:137-177.  dists/knns as returned by the search, queries already normalised."""
    dists = torch.as_tensor(dists)
    if metric_type == "do_not_recomp_l2":
        return -1 * dists
    if metric_type == "do_not_recomp_ip":
        return dists
    knns = np.asarray(knns)
    if metric_type == "l2":
        vecs = torch.from_numpy(np.asarray(keys)[knns].astype(np.float32))
        return -1 * torch.sum((queries[:, None, :] - vecs) ** 2, dim=2)
    if metric_type == "ip":
        vecs = torch.from_numpy(np.asarray(keys)[knns].astype(np.float32))
        if cosine:
            vecs = vecs / (vecs ** 2).sum(-1, keepdims=True).sqrt()
        return (vecs * queries[:, None, :]).sum(dim=-1)
    raise ValueError("Invalid knn similarity function!")


def knn_target_prob(sims, knns, vals, targets, t=1.0):
    """p_knn(target) and recall.  knn_model.py:192-217.

    sims [n,k] f32, knns [n,k] i64 (-1 = padding), vals [N] int, targets [n] i64."""
    sims = torch.as_tensor(sims, dtype=torch.float32).clone()
    knns = np.asarray(knns)
    sims.masked_fill_(torch.from_numpy(knns == -1), MASK_VALUE)          # :193
    probs = torch.softmax(sims / t, dim=-1)                              # :196
    knn_vals = torch.from_numpy(np.asarray(vals)[knns]).long()           # :198  (vals[-1] for pads,
    #                                                                       exactly as numpy does)
    mask = knn_vals == torch.as_tensor(targets).long().unsqueeze(-1)     # :211
    tp = torch.zeros(probs.shape[0], 2).scatter_add_(1, mask.long(), probs)   # :212-213
    return tp[:, 1], mask.sum(-1)                                        # :217


def combine_knn_and_vocab_probs(knn_p, vocab_logp, lmbda):
    """log( (1-l) p_lm + l (p_knn + 1e-10) ) in the reference's float32 logsumexp form.

    sequence_scorer.py:55-68 with knn_probs = log(p_knn + eps) of :121."""
    knn_logp = torch.log(torch.as_tensor(knn_p, dtype=torch.float32) + KNN_EPSILON)
    vocab_logp = torch.as_tensor(vocab_logp, dtype=torch.float32)
    comb = torch.stack([vocab_logp, knn_logp], 0)
    coeffs = torch.ones_like(comb)
    coeffs[0] = np.log(1 - lmbda)
    coeffs[1] = np.log(lmbda)
    return torch.logsumexp(comb + coeffs, dim=0)


def perplexity(score_sum, count):
    """eval_lm.py:325-331: avg_nll (base 2) and ppl."""
    avg_nll = -score_sum / count / math.log(2)
    return avg_nll, 2 ** avg_nll
