"""The eval hot path as one object: gather -> HGT -> adaptive softmax -> kNN interpolation.

``GnnLmEngine.score`` is what one iteration of the reference's hot loop computes between
``gen_timer.start()`` and ``gen_timer.stop()`` (fairseq_cli/eval_lm.py:214-219) *plus* the graph
construction the reference does in its DataLoader workers (token_block_dataset.py:287-331), for a
batch of independent token blocks:

    TokenGraphTransformerDecoder.forward          fairseq/models/transformer.py:943-1009
    AdaptiveSoftmax.get_log_prob + gather         fairseq/modules/adaptive_softmax.py:170-206
    KNNModel.get_knn_prob                         knn/knn_model.py:87-101 (search: on the device with ``knn_index``), 179-217
    combine_knn_and_vocab_probs                   fairseq/sequence_scorer.py:55-68

Everything runs on the current HIP stream through libgnnlm_hip.so; nothing synchronises.
"""
from dataclasses import dataclass
from typing import Optional

import torch

from . import ops
from .adaptive_softmax import AdaptiveSoftmax
from .hgt import HGT, CodeStore, NeighborGraph


@dataclass
class BlockBatch:
    """Inputs of one step: ``n_blocks`` independent blocks of ``T`` tokens (device tensors)."""
    ids: torch.Tensor                       # int64 [n_blocks*T, kg]  rows of neighbors.mmap.{kg}
    tgt_feats: torch.Tensor                 # fp16 / fp32 [n_blocks*T, d]  rows of {split}_dstore/keys.npy
    targets: torch.Tensor                   # int64 [n_blocks*T]
    n_blocks: int
    T: int
    knn_sims: Optional[torch.Tensor] = None  # f32 [n_blocks*T, k]   similarities of the kNN search
    knn_ids: Optional[torch.Tensor] = None   # int64 [n_blocks*T, k] (-1 = padding)
    knn_vals: Optional[torch.Tensor] = None  # int32 [n_blocks*T, k] vals[knn_ids] if already fetched (sharded store)
    fetched_codes: Optional[torch.Tensor] = None
    fetched_valid: Optional[torch.Tensor] = None
    fetched_centres_only: bool = False
    fetched_index: Optional[torch.Tensor] = None


class GnnLmEngine:
    def __init__(self, hgt: HGT, asm: AdaptiveSoftmax, store: CodeStore, left: int, right: int,
                 max_intra_context: int = 0, fetcher=None, fetch_vals: bool = False):
        self.hgt, self.asm, self.store = hgt, asm, store
        self.left, self.right, self.max_intra_context = left, right, max_intra_context
        # range-sharded store (dist.ShardedFetcher): the code rows -- and, with fetch_vals, the labels of the kNN ids -- are
        # fetched from their owners inside the step (batches that bring their own fetched_* / knn_vals keep them)
        self.fetcher, self.fetch_vals = fetcher, fetch_vals

    def features(self, batch: BlockBatch) -> torch.Tensor:
        """gcn_feat: HGT output for every token [n_blocks*T, d] (transformer.py:997)."""
        tgt = batch.tgt_feats
        if tgt.dtype == torch.float16:
            tgt = ops.half_to_float(tgt.contiguous())           # token_block_dataset.py:328
        G = NeighborGraph(ids=batch.ids, n_blocks=batch.n_blocks, T=batch.T, left=self.left, right=self.right,
                          store=self.store, fetched_codes=batch.fetched_codes, fetched_valid=batch.fetched_valid,
                          fetched_centres_only=batch.fetched_centres_only, fetched_index=batch.fetched_index,
                          max_intra_context=self.max_intra_context, fetcher=self.fetcher if batch.fetched_codes is None else None)
        return self.hgt(G, features={"tgt": tgt})["tgt"]

    def score(self, batch: BlockBatch, lmbda: float = 0.0, temperature: float = 1.0, knn_index=None, k: int = 0):
        """Per-token log-probabilities.  Returns dict(gcn_feat, lm_logp, logp[, p_knn, recall]).

        ``knn_index`` (an ``ivfpq.IVFPQIndex`` with the labels attached): the kNN search of the step's own queries -- the
        L2-normalised gcn_feat rows, knn_model.py:100,181-184 -- runs on the device inside the step, as it runs inside the
        reference's timer (fairseq_cli/eval_lm.py:214-219 around sequence_scorer.py:115-120); the batch's ``knn_*`` fields are
        then not read.  The softmax is enqueued between the search and the host's one look at its survivor counts."""
        return self.score_finish(self.score_begin(batch, lmbda, temperature, knn_index, k))

    def score_begin(self, batch: BlockBatch, lmbda: float = 0.0, temperature: float = 1.0, knn_index=None, k: int = 0):
        """Enqueue the step up to the search's host read (features, search, softmax) and return a handle for ``score_finish``:
        several batches can be in flight (one per stream), the host looks at a search's survivor counts only when it comes back
        to that batch."""
        x = self.features(batch)
        pending = None
        if lmbda > 0.0 and knn_index is not None:
            qn = x / (x ** 2).sum(-1, keepdim=True).sqrt()
            pending = knn_index.search_begin(qn.contiguous(), k, return_vals=True)
        lm_logp = self.asm.target_log_prob(x, batch.targets)
        return batch, lmbda, temperature, x, lm_logp, pending

    def score_finish(self, handle):
        batch, lmbda, temperature, x, lm_logp, pending = handle
        out = {"gcn_feat": x, "lm_logp": lm_logp, "logp": lm_logp}
        if pending is not None:
            sims, ids, vals = pending.result()
            logp, p_knn, recall = ops.knn_interp(lm_logp, sims, ids, batch.targets, temperature, lmbda,
                                                 n_store=self.store.n_store, knn_vals=vals)
            out.update(logp=logp, p_knn=p_knn, recall=recall, knn_sims=sims, knn_ids=ids, knn_vals=vals)
        elif lmbda > 0.0:                                        # sequence_scorer.py:102
            if batch.knn_sims is None or batch.knn_ids is None:
                raise ValueError("lmbda > 0 needs knn_sims / knn_ids (results of the kNN search)")
            knn_vals = batch.knn_vals
            if knn_vals is None and self.fetcher is not None and self.fetch_vals:
                knn_vals = self.fetcher.fetch_knn_vals(batch.knn_ids)
            logp, p_knn, recall = ops.knn_interp(
                lm_logp, batch.knn_sims, batch.knn_ids, batch.targets, temperature, lmbda,
                vals=self.store.vals, n_store=self.store.n_store,
                row0=getattr(self.store, "vals_row0", self.store.row0), knn_vals=knn_vals)
            out.update(logp=logp, p_knn=p_knn, recall=recall)
        return out
