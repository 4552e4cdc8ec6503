#!/usr/bin/env python3
"""Debug: round-2 candidate sets of the int8-filter path vs the one-pass float32 scan on the synthetic index."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gnnlm_amd.ivfpq import IVFPQIndex
from gnnlm_amd.synthetic import synthetic_ivfpq_index

dev = torch.device("cuda:0")
N = int(os.environ.get("N", 20_000_000)); n = int(os.environ.get("NQ", 1024)); k = 1024
a = synthetic_ivfpq_index(N, 1024, 4096, 64, dev)
b = IVFPQIndex(a.R, a.coarse, a.pq, a.list_off, a.list_ids, a.list_codes, nprobe=32, scan="f32")
a.keep_candidates = b.keep_candidates = True
torch.manual_seed(0)
q = torch.randn(n, 1024, device=dev); q = q / q.norm(dim=1, keepdim=True)
va, ia = a.search_device(q, k, query_block=n)
vb, ib = b.search_device(q, k, query_block=n)
print("identical results:", torch.equal(va, vb), torch.equal(ia, ib))
ca, cb = a.last_candidates, b.last_candidates
print("tau equal:", torch.equal(ca[3], cb[3]))
na, nb = ca[2].cpu().numpy(), cb[2].cpu().numpy()
print("candidates mfma", na.sum(), "f32", nb.sum(), "queries that differ", (na != nb).sum())
bad = np.nonzero(na != nb)[0][:5]
row_of = {}
for r in bad:
    sa = dict(zip(ca[1][r, :na[r]].cpu().numpy().tolist(), ca[0][r, :na[r]].cpu().numpy().tolist()))
    sb = dict(zip(cb[1][r, :nb[r]].cpu().numpy().tolist(), cb[0][r, :nb[r]].cpu().numpy().tolist()))
    miss = [i for i in sb if i not in sa]
    extra = [i for i in sa if i not in sb]
    tau = float(ca[3][r])
    print(f"query {r}: mfma {na[r]} f32 {nb[r]} missing {len(miss)} extra {len(extra)} tau {tau:.6f}")
    lo = a.list_off.cpu().numpy()
    for i in miss[:8]:
        l = np.searchsorted(lo, i, side="right") - 1          # synthetic index: id == row
        print(f"   missing id {i} score {sb[i]:.6f} (score - tau {sb[i] - tau:.2e}) list {l} row-in-list {i - lo[l]} of {lo[l + 1] - lo[l]}  row%16={i % 16} tile-in-list {(i >> 4) - (lo[l] >> 4)}")
