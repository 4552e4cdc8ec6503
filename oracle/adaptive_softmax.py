"""Oracle: tied adaptive-softmax target log-probability (TEST INFRASTRUCTURE).

Restates ``AdaptiveSoftmax.get_log_prob`` with a target
(fairseq/modules/adaptive_softmax.py:170-206, head :24-47, tail :91-115) followed by
``gather_target_probs`` (fairseq/sequence_scorer.py:48-53,89), for the tied configuration of
``transformer_lm_wiki103``: only the target's entry of the dense
``[T, V]`` tensor is ever read, so only that entry is produced.

Weights:
    emb[i]   [size_i, dim_i]   embed_tokens.embeddings.{i}.0.weight  (tied word matrices)
    proj[i]  [d, dim_i]        embed_tokens.embeddings.{i}.1.weight  (i >= 1; used transposed,
                               TiedLinear(transpose=True), :99-101), None for band 0 when
                               dim_0 == d (:31-35)
    class_proj [n_tail, d]     adaptive_softmax.head.class_proj.weight
"""
import torch


def init_adaptive_weights(vocab, d, cutoff, factor=4, seed=0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    cut = list(cutoff) + ([vocab] if vocab > cutoff[-1] else [])
    emb, proj = [], []
    prev = 0
    for i, c in enumerate(cut):
        dim = int(d // (factor ** i))
        size = c - prev
        emb.append((torch.randn(size, dim, generator=g, dtype=torch.float64) * dim ** -0.5).to(dtype))
        proj.append(None if i == 0 else
                    (torch.randn(d, dim, generator=g, dtype=torch.float64) * d ** -0.5).to(dtype))
        prev = c
    class_proj = (torch.randn(len(cut) - 1, d, generator=g, dtype=torch.float64) * d ** -0.5).to(dtype)
    return {"cutoff": cut, "emb": emb, "proj": proj, "class_proj": class_proj}


def target_log_prob(x, target, w):
    """log p(target | x) under the tied adaptive softmax.  x [n,d], target [n] int64 -> [n].

    Band 0:   lsm_head[y]
    Band i>0: lsm_head[cutoff0 + i - 1] + lsm_tail_{i-1}[y - cutoff[i-1]]
    (adaptive_softmax.py:184-203)."""
    cut = w["cutoff"]
    head_w = torch.cat([w["emb"][0], w["class_proj"]], 0)               # [cut0 + n_tail, d]
    head = torch.log_softmax(x @ head_w.t(), dim=1)                     # :184-188
    out = torch.empty(x.shape[0], dtype=x.dtype)
    in0 = target < cut[0]
    out[in0] = head[in0].gather(1, target[in0, None])[:, 0]
    for i in range(1, len(cut)):
        m = (target >= cut[i - 1]) & (target < cut[i])                  # adapt_target :122-145
        if not m.any():
            continue
        xi = x[m] @ w["proj"][i]                                        # TiedLinear(transpose=True)
        tail = torch.log_softmax(xi @ w["emb"][i].t(), dim=1)           # :199-203
        out[m] = head[m, cut[0] + i - 1] + tail.gather(1, (target[m] - cut[i - 1])[:, None])[:, 0]
    return out


def dense_log_prob(x, w):
    """Full [n, V] log-probabilities (get_log_prob with target=None, :189-196) -- test helper."""
    cut = w["cutoff"]
    head_w = torch.cat([w["emb"][0], w["class_proj"]], 0)
    head = torch.log_softmax(x @ head_w.t(), dim=1)
    parts = [head[:, :cut[0]]]
    for i in range(1, len(cut)):
        tail = torch.log_softmax((x @ w["proj"][i]) @ w["emb"][i].t(), dim=1)
        parts.append(tail + head[:, cut[0] + i - 1, None])
    return torch.cat(parts, 1)
