"""The dense-source launches of the star attention (layers >= 1 of a multi-layer model: `star_attn_kernel<4>`, csrc/attn.hip) on their own:
4096 tokens x 128 neighbour rows of 4 KiB gathered through an index (the cross-batch cache's slots), checked against torch.

Round 4 A/B with this tool: the kernel's lane-to-row mapping (lane l owns 64 CONTIGUOUS bytes of a row, four 16-B loads) against an interleaved one
(piece t * 64 + l: every load instruction of a wave covers 1 KiB of contiguous bytes): 1750 us against 2429 us -- a lane's four loads of one 64-byte
stretch merge in the L1, a wave's interleaved loads of one row do not.  The shipped mapping stays."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from gnnlm_amd import _lib
dev = torch.device("cuda:0")
T, kg, H, D = 4096, 128, 8, 1024
g = torch.Generator(device=dev); g.manual_seed(0)
G = 500000
X = torch.randn(G, D, generator=g, device=dev)
U = torch.randn(T, H, D, generator=g, device=dev) * 0.05
ids = torch.randint(0, 1000000, (T, kg), generator=g, device=dev)
xi = torch.randint(0, G, (T * kg,), generator=g, device=dev).to(torch.int32)
Z = torch.empty(T, H, D, device=dev); has = torch.empty(T, device=dev)
a = _lib.gnnlm_star_attn_t()
a.U, a.ids, a.T, a.H, a.D, a.kg = U.data_ptr(), ids.data_ptr(), T, H, D, kg
a.Z, a.has_nb, a.n_store = Z.data_ptr(), has.data_ptr(), 1000000
a.X, a.ldx, a.x_group_stride, a.x_index = X.data_ptr(), D, 1, xi.data_ptr()
for _ in range(3): _lib.call_desc("gnnlm_star_attn", a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): _lib.call_desc("gnnlm_star_attn", a)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
print(f"dense-source star attention, {T} tokens x {kg} neighbours, d = {D}: {us:.1f} us  ({2 * T * kg * D * 4 / us / 1e3:.1f} GB/s of row reads)")
# reference value
xr = X[xi.long()].view(T, kg, D)
s = torch.einsum("tkd,thd->thk", xr, U)
al = torch.softmax(s, dim=2)
ref = torch.einsum("thk,tkd->thd", al, xr)
print("max abs err vs torch:", float((Z - ref).abs().max()))
