#!/usr/bin/env python3
"""Epilogue cost of the f32 GEMM at the HGT projection shape: plain / +bias / +residual / +gate (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnnlm_amd import _lib

dev = torch.device("cuda:0")
M, N, K = 8192, 1024, 1024
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
R = torch.randn(M, N, device=dev); bias = torch.randn(N, device=dev); gate = torch.ones(M, device=dev)
for name, use_b, use_r, use_g, alpha in [("plain", 0, 0, 0, 1.0), ("bias", 1, 0, 0, 1.0), ("bias+residual", 1, 1, 0, 0.5),
                                          ("bias+residual+gate", 1, 1, 1, 1.0), ("residual only", 0, 1, 0, 1.0)]:
    g = _lib.gnnlm_gemm_t()
    g.A, g.lda, g.W, g.ldw, g.C, g.ldc = A.data_ptr(), K, W.data_ptr(), K, C.data_ptr(), N
    g.M, g.N, g.K, g.alpha = M, N, K, alpha
    if use_b: g.bias, g.bias_mode = bias.data_ptr(), 1
    if use_r: g.R, g.ldr = R.data_ptr(), N
    if use_g: g.gate = gate.data_ptr()
    for _ in range(3): _lib.call_desc("gnnlm_gemm_nt", g)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): _lib.call_desc("gnnlm_gemm_nt", g)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print(f"{name:22s} {us:8.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s")
