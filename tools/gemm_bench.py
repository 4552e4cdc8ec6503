#!/usr/bin/env python3
"""Micro-benchmark of the f32 MFMA GEMM on the shapes of one eval step (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnnlm_amd import _lib, ops

dev = torch.device("cuda:0")
SHAPES = [  # name, M, N, K, batch1, batch2
    ("tgt_proj 2048x1024x1024", 2048, 1024, 1024, 1, 1),
    ("vt 1024x256x1024 b8", 1024, 256, 1024, 8, 1),
    ("scores 256x256x128 b64", 256, 256, 128, 8, 8),
    ("pv 256x128x256 b64", 256, 128, 256, 8, 8),
    ("U 2048x1024x128 b8", 2048, 1024, 128, 8, 1),
    ("Zvz 2048x128x1024 b8", 2048, 128, 1024, 8, 1),
    ("head 2048x20002x1024", 2048, 20002, 1024, 1, 1),
    ("tail1 256x40000x256", 256, 40000, 256, 1, 1),
    ("tail2 256x207744x64", 256, 207744, 64, 1, 1),
    ("ntgt 163840x1024x1024", 163840, 1024, 1024, 1, 1),
    ("sq 4096^3", 4096, 4096, 4096, 1, 1),
    ("head8k 8192x20002x1024", 8192, 20002, 1024, 1, 1),
    ("step8k proj 8192x1024x1024", 8192, 1024, 1024, 1, 1),
    ("step8k qkv 8192x3072x1024", 8192, 3072, 1024, 1, 1),
    ("step8k U 8192x1024x128 b8", 8192, 1024, 128, 8, 1),
    ("step8k Zvz 8192x128x1024 b8", 8192, 128, 1024, 8, 1),
]
ORDER = int(os.environ.get("TILE_ORDER", "0"))
PREC = int(os.environ.get("PRECISION", "0"))
sel = sys.argv[1] if len(sys.argv) > 1 else ""
for name, M, N, K, b1, b2 in [s_ for s_ in SHAPES if sel in s_[0]]:
    nb = b1 * b2
    A = torch.randn(nb * M, K, device=dev)
    W = torch.randn(nb * N, K, device=dev)
    C = torch.empty(nb * M, N, device=dev)
    g = _lib.gnnlm_gemm_t()
    g.A, g.lda, g.W, g.ldw, g.C, g.ldc = A.data_ptr(), (0 if os.environ.get("LDA0") else K), W.data_ptr(), K, C.data_ptr(), (0 if os.environ.get("LDC0") else N)   # LDC0=1: every row is stored to row 0 (no HBM write stream: what the stores cost); LDA0=1: every row reads row 0 (a cache-resident A panel: what streaming A costs)
    g.M, g.N, g.K, g.batch1, g.batch2 = M, N, K, b1, b2
    g.tile_order = ORDER
    g.precision = PREC
    g.sA1, g.sA2, g.sW1, g.sW2, g.sC1, g.sC2 = b2 * M * K, M * K, b2 * N * K, N * K, b2 * M * N, M * N
    for _ in range(3):
        _lib.call_desc("gnnlm_gemm_nt", g)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        _lib.call_desc("gnnlm_gemm_nt", g)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    _lib.profile_begin()
    _lib.call_desc("gnnlm_gemm_nt", g)
    torch.cuda.synchronize()
    pr = _lib.profile_end()
    parts = "  ".join(f"{k_.split('_kernel')[0]} {v['total_ms'] * 1e3:.0f}us" for k_, v in pr.items())
    print(f"{name:28s} {us:9.1f} us  {2.0 * M * N * K * nb / us / 1e6:7.1f} TFLOP/s   [{parts}]")
