#!/usr/bin/env python3
"""Cycle breakdown of star_attn_tab_kernel from a -DGNNLM_STAB_EXP=9 build (GNNLM_LIB=.../libstab9.so)."""
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
c = has.view(torch.int32)[: 256 * 16].view(256, 16).double().cpu()
names = ["staging", "pass1", "softmax", "pass2"] + [f"p{p_}_{ph}_{w}" for p_ in (1, 2) for ph in ("L", "M") for w in ("body", "ldswait", "barrier")]
names[14] = "p1_gap_M_to_L"; names[15] = "p1_gap_L_to_M"
for i, n in enumerate(names):
    print(f"{n:12s} mean {c[:, i].mean():10.0f}  min {c[:, i].min():10.0f}  max {c[:, i].max():10.0f}")
print("total", c[:, :4].sum(1).mean())
