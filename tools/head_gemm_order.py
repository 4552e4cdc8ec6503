#!/usr/bin/env python3
"""The softmax head's GEMM (8192 x 20004 x 1024, log-sum-exp epilogue: logits never written) under different tile walks:
    TILE_ORDER=0 (auto) | 1 (n fastest) | 2 (m fastest) | 2 + GM (bands of GM m-tiles)
Prints time / TFLOP/s; under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` the counter gives the L2 -> fabric traffic per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnnlm_amd import _lib

dev = torch.device("cuda:0")
M, N, K = int(os.environ.get("HEAD_M", "8192")), 20004, 1024
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.03
part = torch.empty(M * 4 * ((N + 127) // 128), device=dev)
pick = torch.randint(0, N, (M,), device=dev, dtype=torch.int32)
picked = torch.empty(M, device=dev)
for order in [int(v) for v in os.environ.get("TILE_ORDERS", "0,1,2,4,6,10,18").split(",")]:
    g = _lib.gnnlm_gemm_t()
    g.A, g.lda, g.W, g.ldw = A.data_ptr(), K, W.data_ptr(), K
    g.lse_part, g.lse_pick, g.lse_picked = part.data_ptr(), pick.data_ptr(), picked.data_ptr()
    g.M, g.N, g.K = M, N, K
    g.tile_order = order
    for _ in range(3):
        _lib.call_desc("gnnlm_gemm_nt", g)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = int(os.environ.get("REPS", "20"))
    e0.record()
    for _ in range(reps):
        _lib.call_desc("gnnlm_gemm_nt", g)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"tile_order {order:3d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s", flush=True)
