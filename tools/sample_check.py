#!/usr/bin/env python3
"""The sampled threshold pass on CLUSTERED keys (the 8.4 M-key index of tests/test_ivfpq_mfma_gpu.py::reference_shape_index, sampling
forced although its lists are short): how many queries fail the verification, and is the result the exact one?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnnlm_amd.ivfpq import IVFPQIndex

dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(77)
N, d, nc = 8_400_000, 1024, 3000
centres = torch.randn(nc, d, generator=g, device=dev)
p = 1.0 / torch.arange(1, nc + 1, device=dev, dtype=torch.float64) ** 0.7
which = torch.multinomial(p / p.sum(), N, replacement=True, generator=g)
keys = torch.empty(N, d, device=dev, dtype=torch.float16)
for s in range(0, N, 1 << 19):
    w = which[s:s + (1 << 19)]
    keys[s:s + (1 << 19)] = (centres[w] + 0.8 * torch.randn(w.shape[0], d, generator=g, device=dev)).to(torch.float16)
index = IVFPQIndex.build(keys, 4096, 64, device=dev, cosine=True, nprobe=32, iters=6, seed=5)
nq = 4096
qi = torch.randint(0, nc, (nq,), generator=g, device=dev)
q = centres[qi] + 0.8 * torch.randn(nq, d, generator=g, device=dev)
q = q / q.norm(dim=1, keepdim=True)
for k in (1024, 256, 64):
    index.threshold_sample = 1
    v0, i0 = index.search_device(q, k)
    v0, i0 = v0.clone(), i0.clone()
    for S in (2, 4, 8):
        index.threshold_sample, index.sample_min_keys_per_k, index.sample_fail_frac = S, 0, 1.0
        v, i = index.search_device(q, k)
        st = index.stats
        print(f"k={k} every {S}th tile: failing queries {float(st['underflow']):.0f} of {nq}, searched again {int(st['requeried'])}, "
              f"survivors/query {float(st['survivors']) / nq:.0f}, identical to the exact pass: {bool(torch.equal(v, v0) and torch.equal(i, i0))}")
