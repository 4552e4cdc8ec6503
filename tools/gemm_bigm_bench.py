import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from gnnlm_amd import _lib
dev = torch.device("cuda:0")
for M in (163840, 327680, 655360):
    N = K = 1024
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev); bias = torch.randn(N, device=dev)
    g = _lib.gnnlm_gemm_t()
    g.A, g.lda, g.W, g.ldw, g.C, g.ldc = A.data_ptr(), K, W.data_ptr(), K, C.data_ptr(), N
    g.bias, g.bias_mode = bias.data_ptr(), 1
    g.M, g.N, g.K = M, N, K
    for _ in range(2): _lib.call_desc("gnnlm_gemm_nt", g)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): _lib.call_desc("gnnlm_gemm_nt", g)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 200
    print(f"M={M}: {us:9.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")
    del A, C
