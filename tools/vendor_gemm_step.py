import torch
dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = False
def t(f):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 100
for name, M, N, K in [("qkv", 8192, 3072, 1024), ("proj", 8192, 1024, 1024), ("tail1", 720, 40000, 256), ("tail2", 980, 207744, 64)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
    us = t(lambda: A @ W.t())
    print(f"{name:8s} {M}x{N}x{K}: {us:9.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")
# U: batch 8, [8192 x 128] x [128 x 1024]
q = torch.randn(8, 8192, 128, device=dev); W = torch.randn(8, 1024, 128, device=dev)
us = t(lambda: torch.bmm(q, W.transpose(1, 2)))
print(f"U bmm 8x8192x1024x128: {us:9.1f} us  {2.0 * 8 * 8192 * 1024 * 128 / us / 1e6:7.1f} TFLOP/s")
# as in the step: q is [8192, 8, 128] strided (heads interleaved), out [8192, 8, 1024]
q2 = torch.randn(8192, 8, 128, device=dev)
us = t(lambda: torch.einsum("thk,hnk->thn", q2, W))
print(f"U einsum thk,hnk->thn: {us:9.1f} us")
