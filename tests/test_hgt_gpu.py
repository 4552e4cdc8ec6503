"""HGT forward of the HIP path (gnnlm_hgt_forward through the host mirror) against
  (a) the golden vectors produced by the reference's own HGT/HGTLayer.forward on graphs built by the
      reference's own new_build_graph (tests/golden/make_golden.py), and
  (b) the CPU oracle evaluated in float64 on seeded inputs, up to the WikiText-103 shapes.
GPU only.  Tolerance: float32 path vs float64 truth, |err| <= 5e-5 absolute on LayerNorm-ed outputs
(O(1) magnitude); north_star's bar is |d logp| <= 1e-4 downstream."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import graph as og
from oracle import hgt as ohgt
from oracle import pq as opq

TOL = 5e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def make_store(dev, codes, cen, A, b, vals=None):
    from gnnlm_amd.hgt import CodeStore
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return CodeStore(codes=t(codes), centroids=t(cen), n_store=codes.shape[0], vals=t(vals), A=t(A), b=t(b))


def run_hip(dev, sd, n_layers, n_heads, d, store, nb_ids, n_blocks, T, l, r, tgt, return_ntgt=True):
    from gnnlm_amd.hgt import HGT, NeighborGraph
    model = HGT(in_dim=d, hidden_dim=d, out_dim=d, n_layers=n_layers, n_heads=n_heads)
    missing = model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=True)
    G = NeighborGraph(ids=torch.from_numpy(nb_ids).to(dev), n_blocks=n_blocks, T=T, left=l, right=r, store=store)
    out = model(G, features={"tgt": torch.from_numpy(tgt).to(dev)}, return_ntgt=return_ntgt)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


def _lr(tag):
    return int(tag[tag.index("l") + 1]), int(tag[tag.index("r") + 1])


def test_hgt_golden_reference_forward(dev, golden):
    g, gg = golden("hgt"), golden("graph")
    cases = sorted({k.split(".tgt_in")[0] for k in g.files if k.endswith(".tgt_in")})
    assert len(cases) >= 8
    store = make_store(dev, g["codes"], g["cen"], g["A"], g["b"])
    for key in cases:
        tag, cfg = key.split(".")
        L, H = int(cfg[1]), int(cfg[3:])
        l, r = _lr(tag)
        nb = gg[tag + ".nb"]
        sd = {k[len(key) + 4:]: g[k] for k in g.files if k.startswith(key + ".sd.")}
        out = run_hip(dev, sd, L, H, 32, store, nb, 1, nb.shape[0], l, r, g[key + ".tgt_in"])
        np.testing.assert_allclose(out["tgt"], g[key + f".tgt_out{L - 1}"], atol=TOL, rtol=1e-4, err_msg=key)
        np.testing.assert_allclose(out["ntgt"], g[key + f".ntgt_out{L - 1}"], atol=TOL, rtol=1e-4, err_msg=key)


@pytest.mark.parametrize("L", [1, 2, 3])
def test_block_without_any_neighbour_matches_oracle(dev, golden, L):
    """Every neighbour id -1 (tests/golden/graph_empty.npz: the reference raises before it reaches DGL, so no reference output
    exists): the HIP path scores the block from the causal edges alone, exactly as the float64 oracle on the graph without
    ntgt nodes -- merged, un-merged and with the centre-state cache."""
    g = golden("graph_empty")
    from gnnlm_amd.hgt import HGT, NeighborGraph
    d, H, M, dsub = 32, 2, 4, 4
    rs = np.random.RandomState(9)
    cen = (rs.randn(M, 256, dsub) * 0.5).astype(np.float32)
    A = (rs.randn(M * dsub, d) / 4).astype(np.float32)
    b = (rs.randn(M * dsub) * 0.1).astype(np.float32)
    codes = g["codes"]
    for tag in ("T6k3l2r2", "T4k2l0r0"):
        l, r = _lr(tag)
        nb = g[tag + ".nb"]
        T = nb.shape[0]
        tgt = rs.randn(T, d).astype(np.float16).astype(np.float32)
        sd = {k: v.numpy() for k, v in ohgt.init_hgt_weights(L, d, H, seed=11).items()}
        ref = oracle_hgt(sd, L, H, tgt, nb, codes, cen, A, b, codes.shape[0], l, r)["tgt"].numpy()
        store = make_store(dev, codes, cen, A, b)
        model = HGT(in_dim=d, hidden_dim=d, out_dim=d, n_layers=L, n_heads=H)
        model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=True)
        G = NeighborGraph(ids=torch.from_numpy(nb).to(dev), n_blocks=1, T=T, left=l, right=r, store=store)
        outs = []
        for dedup, slots in ((True, None), (False, None), (True, 64)):
            model.dedup_groups, model.state_cache, model.state_cache_slots = dedup, None, slots
            model.state_cache_gib = 1.0 if slots else 0.0
            outs.append(model(G, features={"tgt": torch.from_numpy(tgt).to(dev)})["tgt"].cpu().numpy())
        assert np.abs(outs[0] - ref).max() < TOL and np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_hgt_adapters_golden(dev, golden):
    """in_dim != hidden_dim != out_dim (hgt.py:476-492,505-513) against the reference's own forward."""
    from gnnlm_amd.hgt import HGT, NeighborGraph
    g = golden("hgt_adapt")
    for key in sorted({k.split(".")[0] for k in g.files if k.endswith(".cfg")}):
        d_in, d_hid, d_out, L, H, T, k, l, r = (int(v) for v in g[key + ".cfg"])
        sd = {kk[len(key) + 4:]: torch.as_tensor(g[kk]) for kk in g.files if kk.startswith(key + ".sd.")}
        model = HGT(in_dim=d_in, hidden_dim=d_hid, out_dim=d_out, n_layers=L, n_heads=H)
        model.load_state_dict(sd, strict=True)
        store = make_store(dev, g["codes"], g["cen"], g[key + ".A"], g[key + ".b"])
        G = NeighborGraph(ids=torch.from_numpy(g[key + ".nb"]).to(dev), n_blocks=1, T=T, left=l, right=r, store=store)
        out = model(G, features={"tgt": torch.from_numpy(g[key + ".tgt_in"]).to(dev)}, return_ntgt=True)
        np.testing.assert_allclose(out["tgt"].cpu().numpy(), g[key + ".tgt_out"], atol=TOL, rtol=1e-4, err_msg=key)
        np.testing.assert_allclose(out["ntgt"].cpu().numpy(), g[key + ".ntgt_out"], atol=TOL, rtol=1e-4, err_msg=key)
        out1 = model(G, features={"tgt": torch.from_numpy(g[key + ".tgt_in"]).to(dev)})        # eval path: tgt only
        assert torch.equal(out1["tgt"], out["tgt"])


def oracle_hgt(sd, L, H, tgt, nb, codes, cen, A, b, n_store, l, r, dtype=torch.float64):
    gr = og.build_graph(nb, np.zeros(nb.shape[0], np.int64), n_store, l, r)
    ntgt = opq.pq_lookup(codes[gr["ntgt_offsets"]], cen).astype(np.float64)
    if A is not None:
        ntgt = (ntgt - (b.astype(np.float64) if b is not None else 0)) @ A.astype(np.float64)
    feats = {"tgt": torch.from_numpy(tgt.astype(np.float64)), "ntgt": torch.from_numpy(ntgt)}
    sdd = {k: torch.as_tensor(v).to(dtype) for k, v in sd.items()}
    return ohgt.hgt_forward(sdd, L, H, feats, gr)


@pytest.mark.parametrize("L", [1, 2, 3])
@pytest.mark.parametrize("cfg", [dict(d=128, H=8, M=16, dsub=8, opq=True, T=24, kg=6, l=2, r=2),
                                 dict(d=64, H=2, M=8, dsub=4, opq=True, T=9, kg=5, l=1, r=0),     # rectangular OPQ
                                 dict(d=64, H=4, M=16, dsub=4, opq=False, T=17, kg=3, l=0, r=3),
                                 dict(d=512, H=8, M=64, dsub=8, opq=True, T=20, kg=12, l=2, r=2)])   # EnWik8-sized model dims
def test_hgt_vs_oracle(dev, L, cfg):
    d, H, M, dsub, T, kg, l, r = (cfg[k] for k in ("d", "H", "M", "dsub", "T", "kg", "l", "r"))
    rs = np.random.RandomState(L * 100 + d)
    n_store, dpq, n_blocks = 500, M * dsub, 2
    codes = rs.randint(0, 256, size=(n_store, M)).astype(np.uint8)
    cen = (rs.randn(M, 256, dsub) * 0.5).astype(np.float32)
    A = (rs.randn(dpq, d) / np.sqrt(dpq)).astype(np.float32) if cfg["opq"] else None
    b = (rs.randn(dpq) * 0.1).astype(np.float32) if cfg["opq"] else None
    sd = {k: v.numpy() for k, v in ohgt.init_hgt_weights(L, d, H, seed=L).items()}
    nb = rs.randint(0, n_store, size=(n_blocks * T, kg)).astype(np.int64)
    nb[rs.rand(*nb.shape) < 0.05] = -1
    nb[3] = -1
    nb[0, 0], nb[1, 0] = 0, n_store - 1
    tgt = rs.randn(n_blocks * T, d).astype(np.float16).astype(np.float32)
    store = make_store(dev, codes, cen, A, b)
    out = run_hip(dev, sd, L, H, d, store, nb, n_blocks, T, l, r, tgt)
    ref_t, ref_n = [], []
    for blk in range(n_blocks):
        sl = slice(blk * T, (blk + 1) * T)
        h = oracle_hgt(sd, L, H, tgt[sl], nb[sl], codes, cen, A, b, n_store, l, r)
        ref_t.append(h["tgt"].numpy())
        ref_n.append(h["ntgt"].numpy())
    assert np.abs(out["tgt"] - np.concatenate(ref_t)).max() < TOL
    assert np.abs(out["ntgt"] - np.concatenate(ref_n)).max() < TOL
    # eval path (no ntgt output requested) gives the same tgt states
    out2 = run_hip(dev, sd, L, H, d, store, nb, n_blocks, T, l, r, tgt, return_ntgt=False)
    assert np.array_equal(out2["tgt"], out["tgt"])


def test_hgt_wikitext103_shape_prefix(dev):
    """Full WikiText-103 block shape (T=256, kg=128, l=r=2, d=1024, H=8, PQ 128x8 + OPQ) on the GPU;
    the causal structure makes the first tokens of the block independent of the rest, so the oracle
    (float64, as written, un-elided) checks the first 6 tokens of a 1- and a 2-layer model."""
    d, H, M, dsub, T, kg, l, r = 1024, 8, 128, 8, 256, 128, 2, 2
    rs = np.random.RandomState(1234)
    n_store = 20000
    codes = rs.randint(0, 256, size=(n_store, M)).astype(np.uint8)
    cen = (rs.randn(M, 256, dsub) * 0.5).astype(np.float32)
    A = (rs.randn(d, d) / np.sqrt(d)).astype(np.float32)
    b = (rs.randn(d) * 0.1).astype(np.float32)
    nb = rs.randint(0, n_store, size=(T, kg)).astype(np.int64)
    nb[rs.rand(T, kg) < 0.001] = -1
    nb[2] = -1
    tgt = rs.randn(T, d).astype(np.float16).astype(np.float32)
    store = make_store(dev, codes, cen, A, b)
    P = 6
    for L in (1, 2):
        sd = {k: v.numpy() for k, v in ohgt.init_hgt_weights(L, d, H, seed=7).items()}
        out = run_hip(dev, sd, L, H, d, store, nb, 1, T, l, r, tgt, return_ntgt=False)
        ref = oracle_hgt(sd, L, H, tgt[:P], nb[:P], codes, cen, A, b, n_store, l, r)["tgt"].numpy()
        err = np.abs(out["tgt"][:P] - ref).max()
        assert err < 1e-4, (L, err)
        assert np.isfinite(out["tgt"]).all()


@pytest.mark.parametrize("precision", [0, 1])
def test_step_is_graph_capturable(dev, precision):
    """The whole step (gather -> HGT -> adaptive softmax -> kNN interpolation) is stream-ordered with device-side
    row counts and no host round trip, so it can be captured in a HIP graph and replayed (bench.py --graph):
    the replay must reproduce the eager results bit for bit, also after the inputs changed in place."""
    from gnnlm_amd.synthetic import build_engine, make_problem, to_batch
    prob = make_problem(n_store=3000, d=64, n_heads=4, M=16, dsub=4, vocab=600, cutoff=[100, 300], T=16, kg=8,
                        left=2, right=2, n_layers=2, k=32, seed=3)
    eng = build_engine(prob, dev)
    eng.hgt.gemm_precision = eng.asm.gemm_precision = precision
    batch = to_batch(prob["block"], dev)
    eager = {k: v.clone() for k, v in eng.score(batch, 0.25, 1.0).items()}
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = eng.score(batch, 0.25, 1.0)
    g.replay()
    torch.cuda.synchronize()
    for k in ("logp", "lm_logp", "gcn_feat"):
        assert torch.equal(out[k], eager[k]), k
    tgt0 = batch.targets.clone()
    batch.targets.copy_(torch.roll(tgt0, 1))                       # same buffers, new contents
    g.replay()
    torch.cuda.synchronize()
    ref = eng.score(batch, 0.25, 1.0)
    assert torch.equal(out["logp"], ref["logp"]) and not torch.equal(out["logp"], eager["logp"])


@pytest.mark.parametrize("L", [2, 3])
def test_hgt_dedup_context_groups(dev, L):
    """Equal context groups of a batch are computed once (the reference's 'todo: merge same nodes',
    token_block_dataset.py:355; a group's states depend on its centre row only): heavy duplicates across tokens and blocks,
    -1 / out-of-store ids, contexts clipped at both ends of the store -- bit-identical to the un-merged run of the same
    kernels, and equal to the un-merged float64 oracle."""
    from gnnlm_amd.hgt import HGT, NeighborGraph
    d, H, M, dsub, T, kg, l, r, nblk = 128, 8, 16, 8, 12, 10, 2, 2, 3
    rs = np.random.RandomState(50 + L)
    n_store = 400
    codes = rs.randint(0, 256, size=(n_store, M)).astype(np.uint8)
    cen = (rs.randn(M, 256, dsub) * 0.5).astype(np.float32)
    A = (rs.randn(M * dsub, d) / np.sqrt(M * dsub)).astype(np.float32)
    b = (rs.randn(M * dsub) * 0.1).astype(np.float32)
    pool = np.concatenate([[0, 1, n_store - 1, n_store - 2], rs.randint(0, n_store, 20)])     # ~24 distinct rows for 360 groups
    nb = pool[rs.randint(0, len(pool), size=(nblk * T, kg))].astype(np.int64)
    nb[rs.rand(*nb.shape) < 0.05] = -1
    nb[2] = -1
    nb[5, 3] = n_store + 4                                                    # not a row of the store
    nb[7] = nb[6]                                                             # identical lists
    tgt = rs.randn(nblk * T, d).astype(np.float16).astype(np.float32)
    sd = {k: v.numpy() for k, v in ohgt.init_hgt_weights(L, d, H, seed=5).items()}
    store = make_store(dev, codes, cen, A, b)
    model = HGT(in_dim=d, hidden_dim=d, out_dim=d, n_layers=L, n_heads=H)
    model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=True)
    G = NeighborGraph(ids=torch.from_numpy(nb).to(dev), n_blocks=nblk, T=T, left=l, right=r, store=store)
    x = torch.from_numpy(tgt).to(dev)
    assert model.dedup_groups
    merged = model(G, features={"tgt": x})["tgt"].cpu().numpy()
    n_all, n_distinct = model.last_groups
    assert n_all == nblk * T * kg and n_distinct <= len(np.unique(pool)) and n_distinct >= 10
    model.dedup_groups = False
    plain = model(G, features={"tgt": x})["tgt"].cpu().numpy()
    assert np.array_equal(merged, plain)                                      # same kernels, same row arithmetic: same bits
    nbo = np.where(nb >= n_store, -1, nb)                                     # (the reference raises IndexError for such a row; here it is no neighbour)
    ref = np.concatenate([oracle_hgt(sd, L, H, tgt[i * T:(i + 1) * T], nbo[i * T:(i + 1) * T], codes, cen, A, b, n_store, l, r)["tgt"].numpy()
                          for i in range(nblk)])
    assert np.abs(merged - ref).max() < 1e-4
    # a batch without a single valid neighbour
    model.dedup_groups = True
    none = NeighborGraph(ids=torch.full((T, kg), -1, dtype=torch.int64, device=dev), n_blocks=1, T=T, left=l, right=r, store=store)
    o1 = model(none, features={"tgt": x[:T]})["tgt"].cpu().numpy()
    assert model.last_groups == (T * kg, 0)
    model.dedup_groups = False
    assert np.array_equal(o1, model(none, features={"tgt": x[:T]})["tgt"].cpu().numpy())


@pytest.mark.parametrize("L", [2, 3])
def test_hgt_row_keyed_layer0_kv(dev, L):
    """Layer 0's K / V of the ntgt slots once per distinct datastore ROW (gnnlm_hgt_io_t.row_table, ABI 11): neighbour lists in RUNS --
    neighbour j of token t + 1 is neighbour j of token t plus one, what consecutive tokens of real kNN-LM retrieval produce -- so
    hardly two centres coincide (nothing for the group-level merge) while neighbouring groups share 4 of their 5 rows; runs that
    walk over both ends of the store, -1 ids, a second call (the row table comes back clean).  Bit-identical to the slot-keyed
    projections (same kernels per row), with and without the centre-state cache, and equal to the un-merged float64 oracle."""
    from gnnlm_amd.hgt import HGT, NeighborGraph
    d, H, M, dsub, T, kg, l, r, nblk = 128, 8, 16, 8, 16, 6, 2, 2, 2
    rs = np.random.RandomState(90 + L)
    n_store = 3000
    codes = rs.randint(0, 256, size=(n_store, M)).astype(np.uint8)
    cen = (rs.randn(M, 256, dsub) * 0.5).astype(np.float32)
    A = (rs.randn(M * dsub, d) / np.sqrt(M * dsub)).astype(np.float32)
    b = (rs.randn(M * dsub) * 0.1).astype(np.float32)
    start = rs.randint(0, n_store - T, size=(nblk, 1, kg))
    start[0, 0, 0], start[0, 0, 1], start[1, 0, 0] = 0, n_store - T, n_store - T - 1          # runs along both ends of the store
    nb = (start + np.arange(T).reshape(1, T, 1)).reshape(nblk * T, kg).astype(np.int64)
    brk = rs.rand(*nb.shape) < 0.1
    nb[brk] = rs.randint(0, n_store, size=int(brk.sum()))                                       # a run broken here and there
    nb[3, 2] = -1
    nb[9] = -1
    tgt = rs.randn(nblk * T, d).astype(np.float16).astype(np.float32)
    sd = {k: v.numpy() for k, v in ohgt.init_hgt_weights(L, d, H, seed=6).items()}
    store = make_store(dev, codes, cen, A, b)
    model = HGT(in_dim=d, hidden_dim=d, out_dim=d, n_layers=L, n_heads=H)
    model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=True)
    model.state_cache_gib = 0.0
    G = NeighborGraph(ids=torch.from_numpy(nb).to(dev), n_blocks=nblk, T=T, left=l, right=r, store=store)
    x = torch.from_numpy(tgt).to(dev)
    assert model.dedup_groups and model.dedup_rows
    by_row = model(G, features={"tgt": x})["tgt"].cpu().numpy()
    n_all, n_distinct = model.last_groups
    slot_rows = (nb[nb >= 0].reshape(-1, 1) + np.arange(-l, r + 1).reshape(1, -1)).reshape(-1)
    slot_rows = slot_rows[(slot_rows >= 0) & (slot_rows < n_store)]
    assert n_distinct > 0.7 * (nb >= 0).sum() and len(np.unique(slot_rows)) < 0.5 * len(slot_rows)    # the regime: distinct centres, shared rows
    again = model(G, features={"tgt": x})["tgt"].cpu().numpy()                # (the row table was handed back clean)
    model.dedup_rows = False
    by_slot = model(G, features={"tgt": x})["tgt"].cpu().numpy()
    assert np.array_equal(by_row, by_slot) and np.array_equal(by_row, again)
    ref = np.concatenate([oracle_hgt(sd, L, H, tgt[i * T:(i + 1) * T], nb[i * T:(i + 1) * T], codes, cen, A, b, n_store, l, r)["tgt"].numpy()
                          for i in range(nblk)])
    assert np.abs(by_row - ref).max() < 1e-4
    # ... and under the centre-state cache (cold, then all hits)
    model.dedup_rows = True
    model.state_cache_gib, model.state_cache_slots, model.state_cache = 1.0, 4096, None
    cold = model(G, features={"tgt": x})["tgt"].cpu().numpy()
    warm = model(G, features={"tgt": x})["tgt"].cpu().numpy()
    assert model.state_cache is not None and np.array_equal(cold, by_row) and np.array_equal(warm, by_row)


@pytest.mark.parametrize("L", [2, 3])
def test_hgt_centre_state_cache(dev, L):
    """The cross-batch cache of context groups' centre states (gnnlm_hgt_io_t.state_cache, ABI 7; slots assigned on the device,
    gnnlm_group_assign, ABI 9): a sequence of batches with overlapping neighbour rows -- cold, partly cached, fully cached,
    generation turnovers (the older half emptied when the one being filled has no room, once with rows of the current batch in
    the half that goes), a batch too large for a generation (falls back to the within-batch merge), new weights (the cache is
    dropped) -- every output bit-identical to the un-cached call, the rows computed per batch those of a host model of the
    two-generation policy, and the result equal to the un-merged float64 oracle."""
    from gnnlm_amd.hgt import HGT, NeighborGraph
    d, H, M, dsub, T, kg, l, r, nblk = 128, 8, 16, 8, 12, 10, 2, 2, 2
    rs = np.random.RandomState(70 + L)
    n_store = 2000
    codes = rs.randint(0, 256, size=(n_store, M)).astype(np.uint8)
    cen = (rs.randn(M, 256, dsub) * 0.5).astype(np.float32)
    A = (rs.randn(M * dsub, d) / np.sqrt(M * dsub)).astype(np.float32)
    b = (rs.randn(M * dsub) * 0.1).astype(np.float32)
    sd = {k: v.numpy() for k, v in ohgt.init_hgt_weights(L, d, H, seed=6).items()}
    store = make_store(dev, codes, cen, A, b)
    model = HGT(in_dim=d, hidden_dim=d, out_dim=d, n_layers=L, n_heads=H)
    model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=True)
    plain_model = HGT(in_dim=d, hidden_dim=d, out_dim=d, n_layers=L, n_heads=H)
    plain_model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=True)
    plain_model.state_cache_gib = 0.0                                         # the within-batch merge only
    cap = 2 * nblk * T * kg                                                   # a generation = the neighbours of one batch (the minimum)
    model.state_cache_slots = cap

    def policy(batches):
        """Host model of the cache's policy: rows computed per batch."""
        caps, gen, halves, out, switches = [cap // 2, cap - cap // 2], 0, [set(), set()], [], 0
        for nb_, _ in batches:
            rows = set(int(v) for v in nb_.reshape(-1) if 0 <= v < n_store)
            miss = rows - halves[0] - halves[1]
            if len(miss) > caps[gen] - len(halves[gen]):
                gen ^= 1
                halves[gen] = set()
                switches += 1
                miss = rows - halves[0] - halves[1]
            halves[gen] |= miss
            out.append(len(miss))
        return out, switches

    def batch(pool_lo, pool_hi, seed, extra=()):
        r_ = np.random.RandomState(seed)
        pool = np.concatenate([[0, n_store - 1], r_.randint(pool_lo, pool_hi, 600), np.asarray(extra, dtype=np.int64)])
        nb = pool[r_.randint(0, len(pool), size=(nblk * T, kg))].astype(np.int64)
        nb[r_.rand(*nb.shape) < 0.05] = -1
        nb[3] = -1
        nb[4, 1] = n_store + 9
        return nb, r_.randn(nblk * T, d).astype(np.float16).astype(np.float32)

    def run(mdl, nb, tgt):
        G = NeighborGraph(ids=torch.from_numpy(nb).to(dev), n_blocks=nblk, T=T, left=l, right=r, store=store)
        return mdl(G, features={"tgt": torch.from_numpy(tgt).to(dev)})["tgt"].cpu().numpy()

    a, b_, c = batch(0, 400, 1), batch(200, 600, 2), batch(600, 1000, 3)
    a_rows = np.unique(a[0][(a[0] >= 0) & (a[0] < n_store)])
    mix = batch(1400, 1800, 4, extra=np.repeat(a_rows[:60], 2))               # new rows + rows of a: a's half is dropped under it
    seq = [a, b_, a, mix, a, c, b_]                                           # 5 generation switches, 2 of them take rows of the batch at hand
    computed = []
    for i, (nb, tgt) in enumerate(seq):
        got = run(model, nb, tgt)
        assert model.state_cache is not None
        computed.append(model.last_groups[1])
        assert np.array_equal(got, run(plain_model, nb, tgt)), i              # same kernels, same per-row arithmetic: same bits
    st = model.state_cache.stats
    want, switches = policy(seq)
    assert computed == want, (computed, want)
    assert computed[1] < plain_model.last_groups[1] and computed[2] == 0      # b: only its new rows; a again: nothing to compute
    assert switches == 5 and computed[3] > 125 and st["generations"] == 1 + switches and st["computed"] == sum(computed)
    nb, tgt = a
    nbo = np.where(nb >= n_store, -1, nb)
    ref = np.concatenate([oracle_hgt(sd, L, H, tgt[i * T:(i + 1) * T], nbo[i * T:(i + 1) * T], codes, cen, A, b, n_store, l, r)["tgt"].numpy()
                          for i in range(nblk)])
    assert np.abs(run(model, nb, tgt) - ref).max() < 1e-4
    # a generation smaller than a batch's neighbour count: the call falls back to the within-batch merge
    model.state_cache_slots, model.state_cache = 6, None
    assert np.array_equal(run(model, *a), run(plain_model, *a)) and model.state_cache.used == 0
    # new weights: cached states are stale and dropped
    model.state_cache_slots, model.state_cache = cap, None
    run(model, *a)
    old_cache = model.state_cache
    assert old_cache.used > 0
    sd2 = {k: v.numpy() for k, v in ohgt.init_hgt_weights(L, d, H, seed=9).items()}
    model.load_state_dict({k: torch.as_tensor(v) for k, v in sd2.items()}, strict=True)
    plain_model.load_state_dict({k: torch.as_tensor(v) for k, v in sd2.items()}, strict=True)
    assert np.array_equal(run(model, *a), run(plain_model, *a)) and model.state_cache is not old_cache


def test_state_cache_bits_at_recipe_shapes(dev):
    """The cached call computes a DIFFERENT number of rows per launch than the un-cached one (a few new groups instead of all of
    them), i.e. its GEMMs may take another kernel (64x64 tiles instead of the hand-scheduled 128x128 loop): the recipe's shapes
    (d = 1024, H = 8, PQ 128x8 + OPQ, k_g = 128, l = r = 2, L = 3, one 256-token block = 32 k groups) with 0 / 90 / 99.7 / 100 % of the
    neighbour rows already cached -- bit-identical to the un-cached call every time."""
    from gnnlm_amd.hgt import HGT, CodeStore, NeighborGraph
    d, H, M, dsub, T, kg, L, N = 1024, 8, 128, 8, 256, 128, 3, 1_000_000
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    codes = torch.randint(0, 256, (N, M), generator=g, device=dev, dtype=torch.uint8)
    cen = torch.randn(M, 256, dsub, generator=g, device=dev) * 0.5
    A = torch.linalg.qr(torch.randn(d, d, generator=g, device=dev))[0].contiguous()
    b = torch.randn(d, generator=g, device=dev) * 0.1
    store = CodeStore(codes=codes, centroids=cen, n_store=N, vals=None, A=A, b=b)
    torch.manual_seed(1)
    cached = HGT(in_dim=d, hidden_dim=d, out_dim=d, n_layers=L, n_heads=H)
    plain = HGT(in_dim=d, hidden_dim=d, out_dim=d, n_layers=L, n_heads=H)
    plain.load_state_dict(cached.state_dict())
    plain.state_cache_gib, cached.state_cache_slots = 0.0, 200000
    ids_a = torch.randint(0, N, (T, kg), generator=g, device=dev)
    x = torch.randn(T, d, generator=g, device=dev)
    run = lambda m, ids: m(NeighborGraph(ids=ids, n_blocks=1, T=T, left=2, right=2, store=store), features={"tgt": x})["tgt"]
    computed = []
    for frac in (0.0, 0.9, 0.997, 1.0):
        ids_b = ids_a.clone()
        fresh = torch.rand(T, kg, generator=g, device=dev) >= frac
        ids_b[fresh] = torch.randint(0, N, (int(fresh.sum()),), generator=g, device=dev)
        cached.state_cache = None
        run(cached, ids_a)                                                    # fills the cache
        got = run(cached, ids_b)
        computed.append(cached.last_groups[1])
        assert torch.equal(got, run(plain, ids_b)), frac
    assert computed[0] > 30000 and 2000 < computed[1] < 5000 and 0 < computed[2] < 300 and computed[3] == 0


def test_group_assign_on_device(dev):
    """gnnlm_group_assign (ABI 9) against numpy: merge mode = np.unique up to the order of the groups (arrival order), the row ->
    group table handed back clean; cache mode = the two-generation policy, slot by slot, over a sequence with switches.  Sizes
    from a handful to half a million ids; -1 / out-of-store ids are not neighbours."""
    from gnnlm_amd.hgt import CentreStateCache, _group_assign
    for n, n_store, n_rows in [(0, 50, 5), (7, 50, 5), (5000, 1000, 300), (524288, 3_000_000, 200_000)]:
        rs = np.random.RandomState(n + 1)
        pool = rs.randint(0, n_store, size=max(n_rows, 1))
        ids = pool[rs.randint(0, len(pool), size=n)].astype(np.int64)
        ids[rs.rand(n) < 0.02] = -1
        ids[rs.rand(n) < 0.01] = n_store + 3
        t = torch.from_numpy(ids).to(dev)
        table = torch.full((n_store,), -1, dtype=torch.int32, device=dev)
        gids, _, gidx, cnt = _group_assign(t, n_store, table)
        torch.cuda.synchronize()
        k = int(cnt[0])
        ok = (ids >= 0) & (ids < n_store)
        want = np.unique(ids[ok])
        got = gids[:k].cpu().numpy()
        assert k == len(want) and np.array_equal(np.sort(got), want)
        assert bool((gids[k:n] == -1).all()) and bool((table == -1).all())
        gi = gidx[:n].cpu().numpy()
        assert (gi[~ok] == -1).all() and (gi[ok] >= 0).all() and (gi[ok] < k).all() and np.array_equal(got[gi[ok]], ids[ok])
    # cache mode
    n_store, n, cap = 5000, 600, 1400
    cache = CentreStateCache(n_store, 2, 8, cap, dev)
    halves, gen, fills = [dict(), dict()], 0, [0, 0]
    rs = np.random.RandomState(9)
    switches = 0
    for step in range(12):
        lo = int(rs.randint(0, 4)) * 1000
        ids = rs.randint(lo, lo + 1500, size=n).astype(np.int64)
        ids[rs.rand(n) < 0.03] = -1
        gids, gslot, gidx, cnt = cache.assign(torch.from_numpy(ids).to(dev))
        torch.cuda.synchronize()
        rows = set(int(v) for v in ids if v >= 0)
        miss = rows - set(halves[0]) - set(halves[1])
        switched = len(miss) > (cap // 2 if gen == 0 else cap - cap // 2) - fills[gen]
        if switched:
            gen ^= 1
            halves[gen], fills[gen] = dict(), 0
            switches += 1
            miss = rows - set(halves[0]) - set(halves[1])
        k = int(cnt[0])
        assert k == len(miss) and int(cnt[1]) == int(switched)
        got_rows, got_slots = gids[:k].cpu().numpy(), gslot[:k].cpu().numpy()
        assert set(got_rows.tolist()) == miss
        base = (0 if gen == 0 else cap // 2) + fills[gen]
        assert sorted(got_slots.tolist()) == list(range(base, base + k))                    # the next free slots of the half being filled
        for r_, s_ in zip(got_rows.tolist(), got_slots.tolist()):
            halves[gen][r_] = s_
        fills[gen] += k
        slot_of = {**halves[0], **halves[1]}
        gi = gidx[:n].cpu().numpy()
        assert all((gi[j] == -1) if ids[j] < 0 else (gi[j] == slot_of[int(ids[j])]) for j in range(n))
        assert cache.id_of_slot[torch.from_numpy(got_slots).long().to(dev)].cpu().tolist() == got_rows.tolist()
    st = cache.stats
    assert switches >= 3 and st["generations"] == 1 + switches and cache.used == fills[0] + fills[1]
    assert cache.assign(torch.zeros(cap // 2 + 1, dtype=torch.int64, device=dev)) is None  # more neighbours than a generation holds


def test_merged_multilayer_step_is_graph_capturable(dev):
    """L = 3 with merging ON inside a HIP graph: the group assignment, the device-side group count every ntgt kernel reads and
    the centre-state cache leave nothing for the host to decide, so the step -- captured on the stream that ran the eager
    warm-up (the persistent tables belong to it) -- replays bit for bit, also after the neighbour ids changed IN PLACE to a
    batch with a different number of distinct groups, with the cache warm, cold, and off."""
    from gnnlm_amd.synthetic import build_engine, make_problem, to_batch
    prob = make_problem(n_store=3000, d=64, n_heads=4, M=16, dsub=4, vocab=600, cutoff=[100, 300], T=16, kg=8,
                        left=2, right=2, n_layers=3, k=32, seed=5)
    for cache_slots in (None, 2048):
        eng = build_engine(prob, dev)
        eng.hgt.state_cache_gib, eng.hgt.state_cache_slots = (0.0, None) if cache_slots is None else (1.0, cache_slots)
        plain = build_engine(prob, dev)
        plain.hgt.dedup_groups = False
        batch = to_batch(prob["block"], dev)
        ids0 = batch.ids.clone()
        ids1 = ids0.clone()
        ids1[:, 1:] = ids1[:, :1]                                      # every token: one distinct neighbour row
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            eng.score(batch, 0.25, 1.0)                                # eager warm-up on the capture stream: tables, cache, workspace
            n_all, n0 = eng.hgt.last_groups
            assert 0 < n0 < n_all
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                out = eng.score(batch, 0.25, 1.0)
            assert eng.hgt._last_groups is not None                    # the captured step is the merged one
            for ids in (ids0, ids1, ids0):
                batch.ids.copy_(ids)
                g.replay()
                s.synchronize()
                got = {k: out[k].clone() for k in ("logp", "gcn_feat")}
                ref = plain.score(batch, 0.25, 1.0)
                for k in got:
                    assert torch.equal(got[k], ref[k]), (cache_slots, k)
        torch.cuda.current_stream().wait_stream(s)


def test_cache_is_cleared_when_a_forward_fails(dev):
    """Cache slots are assigned (on the device) BEFORE the rows' states are computed: a forward that fails in between -- here the
    sharded store's fetch raises -- must not leave slots that later batches would read as hits.  The cache is emptied, the
    error propagates, the next forward computes everything again and equals the un-cached result."""
    from gnnlm_amd.hgt import HGT, NeighborGraph
    d, H, M, dsub, T, kg, L = 64, 4, 8, 8, 8, 6, 2
    rs = np.random.RandomState(3)
    n_store = 300
    codes = rs.randint(0, 256, size=(n_store, M)).astype(np.uint8)
    cen = (rs.randn(M, 256, dsub) * 0.5).astype(np.float32)
    A = (rs.randn(M * dsub, d) / 8).astype(np.float32)
    b = (rs.randn(M * dsub) * 0.1).astype(np.float32)
    store = make_store(dev, codes, cen, A, b)
    sd = ohgt.init_hgt_weights(L, d, H, seed=2)
    cached, plain = (HGT(in_dim=d, hidden_dim=d, out_dim=d, n_layers=L, n_heads=H) for _ in range(2))
    for mdl in (cached, plain):
        mdl.load_state_dict(sd, strict=True)
    plain.state_cache_gib, cached.state_cache_slots = 0.0, 512
    nb = torch.from_numpy(rs.randint(0, n_store, size=(T, kg)).astype(np.int64)).to(dev)
    x = torch.from_numpy(rs.randn(T, d).astype(np.float32)).to(dev)

    class Broken:
        def fetch_groups(self, *a):
            raise RuntimeError("link down")

    ref = plain(NeighborGraph(ids=nb, n_blocks=1, T=T, left=2, right=2, store=store), features={"tgt": x})["tgt"]
    good = NeighborGraph(ids=nb, n_blocks=1, T=T, left=2, right=2, store=store)
    assert torch.equal(cached(good, features={"tgt": x})["tgt"], ref) and cached.state_cache.used > 0
    # (a graph with a fetcher has its own cache: code rows ride beside the states)
    with pytest.raises(RuntimeError, match="link down"):
        cached(NeighborGraph(ids=nb, n_blocks=1, T=T, left=2, right=2, store=store, fetcher=Broken()), features={"tgt": x})
    assert cached.state_cache is not None and cached.state_cache.used == 0           # emptied, not left half-filled
    assert torch.equal(cached(good, features={"tgt": x})["tgt"], ref) and cached.last_groups[1] > 0   # ... and everything computed again
