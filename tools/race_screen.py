#!/usr/bin/env python3
"""Race screen (GPU box): every kernel here is deterministic, so repeated launches on the same inputs must be
bit-identical; a barrier / LDS-DMA ordering bug shows up as a rare differing tile long before it fails a tolerance."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnnlm_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
REPS = int(os.environ.get("REPS", 40))


def screen(name, fn):
    ref = fn()
    ref = [t.clone() for t in (ref if isinstance(ref, (tuple, list)) else (ref,))]
    bad = 0
    for _ in range(REPS):
        out = fn()
        out = out if isinstance(out, (tuple, list)) else (out,)
        bad += any(not torch.equal(a, b) for a, b in zip(out, ref))
    torch.cuda.synchronize()
    print(f"{name:44s} {'OK' if bad == 0 else f'{bad} / {REPS} MISMATCH'}")
    return bad


def R(*s): return torch.randn(*s, generator=g, device=dev)

bad = 0
A, W = R(4000, 128), R(33000, 128)
pick = torch.randint(0, 33000, (4000,), generator=g, device=dev, dtype=torch.int32)
bad += screen("LSE, LDS-DMA 256x256 tiles", lambda: ops.gemm_lse(A, W, pick, alpha=0.07))
A2, W2 = R(1000, 256), R(5000, 256)
pick2 = torch.randint(0, 5000, (1000,), generator=g, device=dev, dtype=torch.int32)
bad += screen("LSE, LDS-DMA 128x128 tiles", lambda: ops.gemm_lse(A2, W2, pick2, alpha=0.1))
A3, W3 = R(900, 64), R(60000, 64)
pick3 = torch.randint(0, 60000, (900,), generator=g, device=dev, dtype=torch.int32)
bad += screen("LSE, A-stationary K = 64", lambda: ops.gemm_lse(A3, W3, pick3, alpha=0.2))
A4, W4, b4, R4 = R(20000, 128), R(6700, 128), R(6700), R(20000, 6700)
bad += screen("store, LDS-DMA 256x256 tiles", lambda: ops.gemm_nt(A4, W4, bias=b4, residual=R4, alpha=0.3))
A5, W5 = R(4100, 288), R(8200, 288)
for prec in ("f32", "bf16x3", "bf16x6"):
    bad += screen(f"store, {prec}, 4100 x 8200 x 288", lambda: ops.gemm_nt(A5, W5, precision=prec))
bad += screen("LSE, bf16x6 planes 256x256", lambda: ops.gemm_lse(A, W, pick, alpha=0.07, precision="bf16x6"))
nb, T, H, dk = 4, 256, 8, 128
Q, K, V = R(nb * T, H * dk) * 0.3, R(nb * T, H * dk) * 0.3, R(nb * T, H * dk)
bad += screen("fused causal attention", lambda: ops.causal_attn(Q, K, V, nb, T, H, 0))
Tn, M, dsub, kg, N = 512, 128, 8, 128, 200000
codes = torch.randint(0, 256, (N, M), generator=g, device=dev, dtype=torch.uint8)
cen = R(M, 256, dsub)
U = R(Tn, H, M * dsub) / 32
ids = torch.randint(0, N, (Tn, kg), generator=g, device=dev)
bad += screen("star attention (wave roles)", lambda: ops.star_attn(U, ids, codes=codes, centroids=cen))
A6, W6, b6 = R(4099, 1024), R(2100, 1024), R(2100)
pick6 = torch.randint(0, 2100, (4099,), generator=g, device=dev, dtype=torch.int32)
bad += screen("store, hand-placed main loop (K = 1024)", lambda: ops.gemm_nt(A6, W6, bias=b6))
bad += screen("LSE, hand-placed main loop (K = 1024)", lambda: ops.gemm_lse(A6, W6, pick6, alpha=0.05))
A7, W7, b7 = R(1500, 1024), R(256, 1024), R(256)
rows7 = torch.randint(0, 1500, (900,), generator=g, device=dev, dtype=torch.int32)
bad += screen("store, narrow output, k split over waves", lambda: ops.gemm_nt(A7, W7, bias=b7, a_rows=rows7))
# IVF-PQ search over the packed image: survivors are appended by atomics in any order, the selected neighbours may not depend on it
from gnnlm_amd.synthetic import synthetic_ivfpq_index
idx = synthetic_ivfpq_index(400_000, 256, 64, 64, dev, nprobe=8)
q = R(300, 256)
q = q / q.norm(dim=1, keepdim=True)
bad += screen("IVF-PQ search, packed scan (M = 64)", lambda: idx.search_device(q, 256))
sys.exit(1 if bad else 0)
