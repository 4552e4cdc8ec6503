"""Determinism / race screen of the barrier- and LDS-DMA-synchronised kernels (tools/race_screen.py): repeated
launches on identical inputs must be bit-identical."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kernels_are_bit_stable():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "race_screen.py")], capture_output=True, text=True,
                       timeout=900, env=dict(os.environ, REPS="12"), cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "MISMATCH" not in p.stdout and p.stdout.count("OK") >= 14


def test_star_attn_tab_repeatable_under_load():
    """The table-resident star kernel synchronises its waves with raw s_barrier / s_waitcnt pairs and hand-issued LDS-DMA:
    300 launches on the same inputs (8192 tokens: eight workgroup rounds per CU, every launch a different interleaving
    of the loader and compute waves) must reproduce the first result bit for bit, with other kernels queued in between."""
    import torch
    from gnnlm_amd import ops
    dev = torch.device("cuda:0")
    T, H, M, dsub, kg, N = 8192, 8, 128, 8, 128, 2_000_000
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    codes = torch.randint(0, 256, (N, M), generator=g, device=dev, dtype=torch.uint8)
    cen = torch.randn(M, 256, dsub, generator=g, device=dev)
    U = torch.randn(T, H, M * dsub, generator=g, device=dev) / 32
    ids = torch.randint(-1, N, (T, kg), generator=g, device=dev)
    Z0, has0 = ops.star_attn(U, ids, codes=codes, centroids=cen)
    Z0, has0 = Z0.clone(), has0.clone()
    noise = torch.randn(4096, 4096, device=dev)
    for it in range(300):
        if it % 7 == 0:
            noise = noise @ noise.t() * 1e-4                      # something else on the device between launches
        Z, has = ops.star_attn(U, ids, codes=codes, centroids=cen)
        if it % 25 == 0 or it == 299:
            assert torch.equal(Z, Z0) and torch.equal(has, has0), it
    torch.cuda.synchronize()


def test_gemm_hand_placed_loop_repeatable_under_load():
    """The GEMM kernel whose k-loop is one inline-asm statement orders its LDS traffic with hand-written s_waitcnt / s_barrier
    (staging registers written by buffer loads the compiler does not see, two LDS buffers, fragment reads a k-group ahead):
    200 launches of a store problem and of a log-sum-exp problem must reproduce the first result bit for bit, with other
    work queued in between -- and agree with the kernels that leave the schedule to the compiler (GNNLM_GEMM_SCHED=0)."""
    import torch
    from gnnlm_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    A = torch.randn(4099, 1024, generator=g, device=dev)
    W = torch.randn(2100, 1024, generator=g, device=dev)
    bias = torch.randn(2100, generator=g, device=dev)
    pick = torch.randint(0, 2100, (4099,), generator=g, device=dev, dtype=torch.int32)
    m_dev = torch.tensor([4000], dtype=torch.int32, device=dev)
    C0 = ops.gemm_nt(A, W, bias=bias).clone()
    l0, p0 = ops.gemm_lse(A, W, pick, alpha=0.05, m_dev=m_dev)
    l0, p0 = l0.clone(), p0.clone()
    noise = torch.randn(2048, 2048, device=dev)
    for it in range(200):
        if it % 5 == 0:
            noise = noise @ noise.t() * 1e-4
        C = ops.gemm_nt(A, W, bias=bias)
        l, p = ops.gemm_lse(A, W, pick, alpha=0.05, m_dev=m_dev)
        if it % 20 == 0 or it == 199:
            assert torch.equal(C, C0), it
            assert torch.equal(l[:4000], l0[:4000]) and torch.equal(p[:4000], p0[:4000]), it
    torch.cuda.synchronize()
    # same problem on the compiler-scheduled kernels (a fresh process: the switch is read once)
    code = ("import torch, sys; sys.path.insert(0, %r); from gnnlm_amd import ops; g = torch.Generator(device='cuda:0'); g.manual_seed(11); "
            "A = torch.randn(4099, 1024, generator=g, device='cuda:0'); W = torch.randn(2100, 1024, generator=g, device='cuda:0'); "
            "bias = torch.randn(2100, generator=g, device='cuda:0'); torch.save(ops.gemm_nt(A, W, bias=bias).cpu(), sys.argv[1])" % ROOT)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "c.pt")
        subprocess.run([sys.executable, "-c", code, f], check=True, env=dict(os.environ, GNNLM_GEMM_SCHED="0"), timeout=600, cwd=ROOT)
        ref = torch.load(f)
    assert (C0.cpu() - ref).abs().max().item() < 2e-3             # same products, another summation order inside a stage


def test_graph_capture_lanes_have_private_workspaces():
    """`eval_lm --graph-capture` with several lanes: every lane's graphs are captured on the lane's own stream, so their HGT / softmax
    workspaces (keyed by the stream) are private -- graphs of different lanes replayed CONCURRENTLY at the recipe's real batch shape
    (one 256-token block, k_g = 128, d = 1024) return, batch by batch, the bits of the eager single-stream forward.  (Captured on
    torch's shared default capture stream they all got the same scratch addresses and raced: ADVICE r05.)"""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from gnnlm_amd.hgt import NeighborGraph
    from gnnlm_amd.model import GnnLmModel
    argv = sys.argv
    sys.argv = ["bench.py", "--n-store", "2000000", "--blocks", "1", "--pool", "12"]
    try:
        args = bench.parse()
    finally:
        sys.argv = argv
    dev = torch.device("cuda:0")
    eng, _, _, _, (d, vocab) = bench.build(args, dev, 0, 1)
    batches = bench.make_batches(args, dev, 0, d, vocab)
    model = GnnLmModel(eng.hgt, eng.asm, None)

    def run(b):
        g = NeighborGraph(ids=b.ids, n_blocks=1, T=b.T, left=2, right=2, store=eng.store, tgt_h=b.tgt_feats)
        tgt = b.targets.view(1, b.T)
        out = model.forward(tgt, graph=g)
        return out[0].clone(), model.target_log_probs(out, tgt).clone()
    ref = [run(b) for b in batches]                                # eager, one stream
    torch.cuda.synchronize()
    model.graph_capture = True
    lanes = [torch.cuda.Stream(device=dev) for _ in range(3)]
    for s_ in lanes:
        s_.wait_stream(torch.cuda.current_stream())
    for rep in range(6):
        got = []
        for i, b in enumerate(batches):
            with torch.cuda.stream(lanes[i % 3]):
                got.append(run(b))
        torch.cuda.synchronize()
        for i, ((x, lp), (x0, lp0)) in enumerate(zip(got, ref)):
            assert torch.equal(x, x0) and torch.equal(lp, lp0), (rep, i)
    assert len({k_[-1] for k_ in model._graphs if k_[0] == "fwd"}) == 3     # one set of graphs per lane stream
