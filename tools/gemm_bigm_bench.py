"""The row-subset projections of the multi-layer path at growing row counts (M x 1024 x 1024, store epilogue, plain and with a
row map): does the rate hold beyond the 655,360 rows of a 4-block step?  One launch over all rows against chunked launches."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from gnnlm_amd import _lib
dev = torch.device("cuda:0")
N = K = 1024
W = torch.randn(N, K, device=dev); bias = torch.randn(N, device=dev)


def run(A, C, M, rows=None, chunk=None):
    def launch():
        for m0 in range(0, M, chunk or M):
            m = min(M, m0 + (chunk or M)) - m0
            g = _lib.gnnlm_gemm_t()
            g.A, g.lda, g.W, g.ldw, g.C, g.ldc = A.data_ptr(), K, W.data_ptr(), K, C.data_ptr(), N
            g.bias, g.bias_mode = bias.data_ptr(), 1
            g.M, g.N, g.K = m, N, K
            if rows is not None:
                g.a_rows = g.c_rows = rows.data_ptr() + 4 * m0
                g.a_rows_bound = A.shape[0]
            else:
                g.A, g.C = A.data_ptr() + 4 * K * m0, C.data_ptr() + 4 * N * m0
            _lib.call_desc("gnnlm_gemm_nt", g)
    for _ in range(2): launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): launch()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 200
    return us, 2.0 * M * N * K / us / 1e6


for M in (163840, 655360, 1310720, 2621440):
    A = torch.randn(M, K, device=dev); C = torch.empty(M, N, device=dev)
    us, tf = run(A, C, M)
    line = f"M={M:8d}: one launch {us:9.1f} us {tf:6.1f} TF"
    if M > 655360:
        us, tf = run(A, C, M, chunk=655360)
        line += f" | chunks of 655360 {us:9.1f} us {tf:6.1f} TF"
    # 3 of every 5 rows (the slots layer 0 of a 3-layer model updates), as a row map
    rows = (torch.arange(M // 5 * 3, device=dev) // 3 * 5 + torch.arange(M // 5 * 3, device=dev) % 3).to(torch.int32)
    us, tf = run(A, C, rows.numel(), rows=rows)
    line += f" | row map 3/5 {tf:6.1f} TF"
    if M > 655360:
        us, tf = run(A, C, rows.numel(), rows=rows, chunk=393216)
        line += f" | row map chunked {tf:6.1f} TF"
    print(line)
    del A, C
