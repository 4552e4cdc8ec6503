"""Tied adaptive softmax, target log-probability only -- mirror of
``AdaptiveSoftmax.get_log_prob(input, target)`` + ``gather_target_probs``
(fairseq/modules/adaptive_softmax.py:170-206, fairseq/sequence_scorer.py:48-53,89).

The reference materialises a dense ``[T, V]`` float tensor (274 MB per 256-token block on
WikiText-103) of which only the target column is read; here only the target's band is computed:
head GEMM + row logsumexp, then per tail band a projection GEMM and a band GEMM restricted (device
side) to the rows whose target lies in that band.
"""
import ctypes
from typing import List, Optional

import torch

from . import _lib


class AdaptiveSoftmax:
    def __init__(self, cutoff: List[int], emb: List[torch.Tensor], proj: List[Optional[torch.Tensor]],
                 class_proj: torch.Tensor, device):
        """cutoff: band upper bounds, last = vocab size.  emb[i] [size_i, dim_i]; proj[i] [d, dim_i]
        (``embed_tokens.embeddings.{i}.1.weight``, None for band 0); class_proj [n_tail, d]."""
        assert len(cutoff) == len(emb) == len(proj) and 1 <= len(cutoff) <= 8
        f = lambda t: t.detach().to(device, torch.float32).contiguous()
        self.cutoff = [int(c) for c in cutoff]
        self.d = emb[0].shape[1]
        self.head_w = f(torch.cat([emb[0].float(), class_proj.float()], 0))      # adaptive_softmax.py:24-47
        # tail dims are padded with zero columns to a multiple of 4 (the GEMM's K granularity): exact
        pad = lambda t: torch.nn.functional.pad(t.float(), (0, (-t.shape[1]) % 4))
        self.emb = [None] + [f(pad(e)) for e in emb[1:]]
        self.proj_t = [None] + [f(pad(p).t()) for p in proj[1:]]                 # TiedLinear(transpose=True), :99-101
        w = _lib.gnnlm_adaptive_softmax_t()
        w.d, w.n_bands = self.d, len(cutoff)
        for i, c in enumerate(self.cutoff):
            w.cutoff[i] = c
        w.head_w = self.head_w.data_ptr()
        for i in range(1, len(cutoff)):
            w.proj_t[i] = self.proj_t[i].data_ptr()
            w.emb[i] = self.emb[i].data_ptr()
            w.dim[i] = self.emb[i].shape[1]
        self._w = w
        self._ws = None
        self.gemm_precision = 0     # 0 exact f32 MFMA | 1 bf16x3 | 2 bf16x6

    @classmethod
    def from_state_dict(cls, sd, cutoff, vocab, device, prefix="decoder."):
        """Tied weights of a reference checkpoint."""
        cut = list(cutoff) + ([vocab] if vocab > cutoff[-1] else [])
        emb = [sd[f"{prefix}embed_tokens.embeddings.{i}.0.weight"] for i in range(len(cut))]
        proj = [None] + [sd[f"{prefix}embed_tokens.embeddings.{i}.1.weight"] for i in range(1, len(cut))]
        return cls(cut, emb, proj, sd[f"{prefix}adaptive_softmax.head.class_proj.weight"], device)

    def release_stream_state(self, keep=()):
        """Free the scratch arenas of every stream but `keep` (raw handles): see HGT.release_stream_state."""
        if self._ws:
            for k_ in [k_ for k_ in self._ws if k_ not in set(keep)]:
                del self._ws[k_]

    def target_log_prob(self, x: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        """x [n, d] f32, target [n] int64 -> log p(target | x) [n]."""
        n = x.shape[0]
        x = x.contiguous()
        target = target.contiguous()
        assert x.dtype == torch.float32 and target.dtype == torch.int64
        out = torch.empty(n, device=x.device, dtype=torch.float32)
        self._w.gemm_precision = self.gemm_precision
        L = _lib.lib()
        need = L.gnnlm_adaptive_workspace_bytes(ctypes.byref(self._w), n)
        key = _lib.raw_stream()                                 # one arena per stream
        if self._ws is None:
            self._ws = {}
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = self._ws[key] = torch.empty(need, device=x.device, dtype=torch.uint8)
        _lib.check(L.gnnlm_adaptive_target_logp(ctypes.byref(self._w), _lib.ptr(x), x.stride(0), _lib.ptr(target), n,
                                                _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.stream()),
                   "gnnlm_adaptive_target_logp")
        return out
