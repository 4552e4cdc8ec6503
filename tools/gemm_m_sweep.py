#!/usr/bin/env python3
"""M x 1024 x 1024 f32 GEMM (store epilogue + bias) over a range of M: this build's kernel against the vendor library's (torch.matmul),
time per row and the marginal rate between successive sizes.  Reference point for the ntgt projections of the multi-layer path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnnlm_amd import _lib
dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = False
N = K = 1024
W = torch.randn(N, K, device=dev); bias = torch.randn(N, device=dev)


def timed(f, reps=8):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


prev = None
for M in [int(x) for x in os.environ.get("MS", "8192,32768,65536,131072,163840,327680,655360").split(",")]:
    A = torch.randn(M, K, device=dev); C = torch.empty(M, N, device=dev)

    def ours():
        g = _lib.gnnlm_gemm_t()
        g.A, g.lda, g.W, g.ldw, g.C, g.ldc = A.data_ptr(), K, W.data_ptr(), K, C.data_ptr(), N
        g.bias, g.bias_mode = bias.data_ptr(), 1
        g.M, g.N, g.K = M, N, K
        if os.environ.get("TILE_ORDER"):
            g.tile_order = int(os.environ["TILE_ORDER"])
        _lib.call_desc("gnnlm_gemm_nt", g)
    uo = timed(ours)
    uv = timed(lambda: torch.addmm(bias, A, W.t()))
    line = f"M={M:7d}: ours {uo:9.1f} us {2.0 * M * N * K / uo / 1e6:6.1f} TF | vendor {uv:9.1f} us {2.0 * M * N * K / uv / 1e6:6.1f} TF"
    if prev:
        line += f" | marginal ours {2.0 * (M - prev[0]) * N * K / (uo - prev[1]) / 1e6:6.1f} TF, vendor {2.0 * (M - prev[0]) * N * K / (uv - prev[2]) / 1e6:6.1f} TF"
    print(line, flush=True)
    prev = (M, uo, uv)
    del A, C
