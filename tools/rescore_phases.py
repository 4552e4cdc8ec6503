#!/usr/bin/env python3
"""Phase times of ivfpq_rescore_kernel's workgroups (a build with -DGNNLM_RESCORE_DBG; GNNLM_LIB=...): thread 0 leaves clock64 deltas in the
last 8 slots of its query's candidate row."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnnlm_amd.synthetic import synthetic_ivfpq_index

dev = torch.device("cuda:0")
idx = synthetic_ivfpq_index(103227021, 1024, 4096, 64, dev)
idx.keep_candidates = True
torch.manual_seed(0)
q = torch.randn(8192, 1024, device=dev); q = q / q.norm(dim=1, keepdim=True)
for _ in range(2):
    idx.search_device(q, 1024); torch.cuda.synchronize()
cv = idx.last_candidates[0]
t = cv[:, -8:-3].contiguous().view(torch.int32).double().mean(0).cpu()
names = ["table copy + first loads", "barrier", "thread 0's records", "wait for the other waves", "payloads"]
print("mean ticks per workgroup:", ", ".join("%s %.0f" % (n, x) for n, x in zip(names, t)), "| sum %.0f" % t.sum())
