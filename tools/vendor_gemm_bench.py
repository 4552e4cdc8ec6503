#!/usr/bin/env python3
"""Reference point only: the vendor library's f32 GEMM (torch.matmul -> hipBLASLt / rocBLAS) on the shapes of
tools/gemm_bench.py.  Not used by the product."""
import torch
dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = False
for name, M, N, K in [("head8k", 8192, 20002, 1024), ("proj", 8192, 1024, 1024), ("ntgt", 163840, 1024, 1024), ("sq", 4096, 4096, 4096)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
    for _ in range(3): C = A @ W.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): C = A @ W.t()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"{name:8s} {M}x{N}x{K}: {us:9.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")
