#!/usr/bin/env python3
"""k-selection at the two shapes of the IVF-PQ search: the final selection (k = 1024 of ~4.4 k candidates per query, ragged rows with
ids) and the probe selection (k = 32 of 4096 coarse scores).  GNNLM_TOPK_MERGE_ONLY=1: the chunk-merge kernel for both (A/B)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnnlm_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
n = 8192
def timed(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
cap, k = 16384, 1024
cv = torch.randn(n, cap, device=dev)
ci = torch.randint(0, 1 << 40, (n, cap), device=dev)
cc = torch.randint(3800, 4900, (n,), device=dev, dtype=torch.int32)
bv = torch.empty(n, k, device=dev); bi = torch.empty(n, k, device=dev, dtype=torch.int64)
print(f"final selection  k={k} of ~4.4k ragged candidates: {timed(lambda: ops.topk_merge(cv, bv, bi, ids=ci, largest=True, init=True, row_ncols=cc)):.3f} ms per {n} rows")
cs = torch.randn(n, 4096, device=dev)
pv = torch.empty(n, 32, device=dev); pi = torch.empty(n, 32, device=dev, dtype=torch.int64)
print(f"probe selection  k=32 of 4096: {timed(lambda: ops.topk_merge(cs, pv, pi, largest=True, init=True)):.3f} ms per {n} rows")
