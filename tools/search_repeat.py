#!/usr/bin/env python3
"""Repeatability of the on-device IVF-PQ search: REPS searches of the same batch must return the same bits (the filter's records arrive in
any order; nothing downstream may depend on it).  SKEW=1.0 for log-normal list lengths."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnnlm_amd.synthetic import synthetic_ivfpq_index

dev = torch.device("cuda:0")
idx = synthetic_ivfpq_index(int(os.environ.get("N", 103227021)), 1024, 4096, 64, dev, skew=float(os.environ.get("SKEW", 0.0)))
torch.manual_seed(0)
q = torch.randn(int(os.environ.get("NQ", 8192)), 1024, device=dev); q = q / q.norm(dim=1, keepdim=True)
v0, i0 = idx.search_device(q, 1024)
v0, i0 = v0.clone(), i0.clone()
bad = 0
reps = int(os.environ.get("REPS", 100))
for r in range(reps):
    v, i = idx.search_device(q, 1024)
    if not (torch.equal(v, v0) and torch.equal(i, i0)):
        bad += 1
print(f"{reps} repeated searches of {q.shape[0]} queries: {bad} differ from the first")
sys.exit(1 if bad else 0)
