#!/usr/bin/env python3
"""Micro-benchmark of the star attention kernels at the WikiText-103 shape (run on the GPU box).
GNNLM_STAR_GENERIC=1 selects the generic (lane-owns-a-table-slice) kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnnlm_amd import ops
dev = torch.device("cuda:0")
T, H, M, dsub, kg, N = int(os.environ.get("T", 2048)), 8, 128, 8, 128, 20_000_000
g = torch.Generator(device=dev); g.manual_seed(0)
codes = torch.randint(0, 256, (N, M), generator=g, device=dev, dtype=torch.uint8)
cen = torch.randn(M, 256, dsub, generator=g, device=dev)
U = torch.randn(T, H, M * dsub, generator=g, device=dev) / 32
ids = torch.randint(0, N, (T, kg), generator=g, device=dev)
for _ in range(3):
    Z, has = ops.star_attn(U, ids, codes=codes, centroids=cen)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    Z, has = ops.star_attn(U, ids, codes=codes, centroids=cen)
e1.record(); torch.cuda.synchronize()
print(f"star_attn T={T}: {e0.elapsed_time(e1) * 100:.1f} us   checksum {Z.double().sum().item():.6f}")
