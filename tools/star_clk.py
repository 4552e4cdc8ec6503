#!/usr/bin/env python3
"""Cycle stamps of star_attn_tab_kernel from a -DGNNLM_STAB_CLK=1 build (GNNLM_LIB=gnn-lm_amd/build/exp/lib<name>.so):
section lengths per workgroup for waves 0 and 4."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnnlm_amd import ops
dev = torch.device("cuda:0")
T, H, M, dsub, kg, N = 8192, 8, 128, 8, 128, 20_000_000
g = torch.Generator(device=dev); g.manual_seed(0)
codes = torch.randint(0, 256, (N, M), generator=g, device=dev, dtype=torch.uint8)
cen = torch.randn(M, 256, dsub, generator=g, device=dev)
U = torch.randn(T, H, M * dsub, generator=g, device=dev) / 32
ids = torch.randint(0, N, (T, kg), generator=g, device=dev)
for _ in range(3):
    Z, has = ops.star_attn(U, ids, codes=codes, centroids=cen)
torch.cuda.synchronize()
c = has.view(torch.int32)[: 256 * 32].view(256, 2, 16).to(torch.int64).cpu()
base = c[:, 0:1, 0:1]
d = ((c - base) & 0xFFFFFFFF).double()          # cycles since wave 0 entered the kernel
for w, nm in ((0, "wave 0"), (1, "wave 4")):
    x = d[:, w]
    print(nm, "sections: staging %.0f  pass1 %.0f  softmax %.0f  pass2 %.0f  total %.0f" % tuple(
        [(x[:, i + 1] - x[:, i]).mean().item() for i in range(4)] + [(x[:, 4] - x[:, 0]).mean().item()]))
raw = has.view(torch.int32)[: 256 * 32].view(256, 32).double().cpu()
print("loader wave 8, cycles per sweep: pass 1 issue %.0f land %.0f barrier %.0f | pass 2 issue %.0f land %.0f barrier %.0f" % tuple(raw[:, 8:14].mean(0).tolist()))
print("staging of wave 0 (one round): ids back %.0f, code pieces back %.0f, LDS written %.0f" % tuple(raw[:, 5:8].mean(0).tolist()))
