"""Edge cases of the HIP path against the oracle: ragged / tiny / empty shapes, --intra-context,
--gcn-context-window, the One-Billion-Word codec shape (OPQ128_512: dsub = 4, rectangular A), k_g = 1024."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import graph as og
from oracle import hgt as ohgt
from oracle import knn as oknn
from oracle import pq as opq
from tests.test_hgt_gpu import make_store, oracle_hgt, run_hip


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("T,kg,l,r,n_blocks", [(1, 1, 0, 0, 1), (5, 3, 2, 2, 3), (7, 2, 0, 1, 2), (33, 4, 1, 1, 1)])
def test_hgt_ragged_shapes(dev, T, kg, l, r, n_blocks):
    d, H, M, dsub, L = 64, 4, 16, 4, 2
    rs = np.random.RandomState(T * 10 + kg)
    n_store = 300
    codes = rs.randint(0, 256, size=(n_store, M)).astype(np.uint8)
    cen = (rs.randn(M, 256, dsub) * 0.5).astype(np.float32)
    sd = {k: v.numpy() for k, v in ohgt.init_hgt_weights(L, d, H, seed=T).items()}
    nb = rs.randint(0, n_store, size=(n_blocks * T, kg)).astype(np.int64)
    tgt = rs.randn(n_blocks * T, d).astype(np.float32)
    store = make_store(dev, codes, cen, None, None)
    out = run_hip(dev, sd, L, H, d, store, nb, n_blocks, T, l, r, tgt)
    ref = np.concatenate([oracle_hgt(sd, L, H, tgt[b * T:(b + 1) * T], nb[b * T:(b + 1) * T], codes, cen, None, None,
                                     n_store, l, r)["tgt"].numpy() for b in range(n_blocks)])
    assert np.abs(out["tgt"] - ref).max() < 5e-5


def test_hgt_empty_batch(dev):
    from gnnlm_amd.hgt import HGT, NeighborGraph
    rs = np.random.RandomState(0)
    store = make_store(dev, rs.randint(0, 256, size=(10, 8)).astype(np.uint8), rs.randn(8, 256, 4).astype(np.float32), None, None)
    model = HGT(in_dim=32, hidden_dim=32, out_dim=32, n_layers=1, n_heads=2)
    G = NeighborGraph(ids=torch.zeros(0, 3, dtype=torch.int64, device=dev), n_blocks=0, T=4, left=1, right=1, store=store)
    out = model(G, features={"tgt": torch.zeros(0, 32, device=dev)})
    assert out["tgt"].shape == (0, 32)


def test_hgt_intra_context(dev):
    """--intra-context c restricts the causal edges to w - u < c (token_block_dataset.py:586-594)."""
    from gnnlm_amd.hgt import HGT, NeighborGraph
    d, H, M, dsub, L, T, kg, ctx = 64, 4, 16, 4, 2, 12, 3, 4
    rs = np.random.RandomState(3)
    n_store = 200
    codes = rs.randint(0, 256, size=(n_store, M)).astype(np.uint8)
    cen = (rs.randn(M, 256, dsub) * 0.5).astype(np.float32)
    sd = ohgt.init_hgt_weights(L, d, H, seed=9)
    nb = rs.randint(0, n_store, size=(T, kg)).astype(np.int64)
    tgt = rs.randn(T, d).astype(np.float32)
    model = HGT(in_dim=d, hidden_dim=d, out_dim=d, n_layers=L, n_heads=H)
    model.load_state_dict(sd)
    G = NeighborGraph(ids=torch.from_numpy(nb).to(dev), n_blocks=1, T=T, left=1, right=1,
                      store=make_store(dev, codes, cen, None, None), max_intra_context=ctx)
    out = model(G, features={"tgt": torch.from_numpy(tgt).to(dev)})["tgt"].cpu().numpy()
    gr = og.build_graph(nb, np.zeros(T, np.int64), n_store, 1, 1, max_intra_context=ctx)
    feats = {"tgt": torch.from_numpy(tgt).double(),
             "ntgt": torch.from_numpy(opq.pq_lookup(codes[gr["ntgt_offsets"]], cen)).double()}
    ref = ohgt.hgt_forward({k: v.double() for k, v in sd.items()}, L, H, feats, gr)["tgt"].numpy()
    assert np.abs(out - ref).max() < 5e-5


def test_hgt_one_billion_word_codec_shape(dev):
    """BASELINE configs[4]: OPQ128_512 -> M = 128, dsub = 4, A [512, 1024], 16 heads."""
    d, H, M, dsub, T, kg, l, r = 1024, 16, 128, 4, 16, 128, 2, 2
    rs = np.random.RandomState(5)
    n_store = 5000
    codes = rs.randint(0, 256, size=(n_store, M)).astype(np.uint8)
    cen = (rs.randn(M, 256, dsub) * 0.5).astype(np.float32)
    A = (rs.randn(M * dsub, d) / np.sqrt(M * dsub)).astype(np.float32)
    b = (rs.randn(M * dsub) * 0.1).astype(np.float32)
    nb = rs.randint(0, n_store, size=(T, kg)).astype(np.int64)
    nb[1] = -1
    tgt = rs.randn(T, d).astype(np.float16).astype(np.float32)
    store = make_store(dev, codes, cen, A, b)
    sd = {k: v.numpy() for k, v in ohgt.init_hgt_weights(2, d, H, seed=2).items()}
    out = run_hip(dev, sd, 2, H, d, store, nb, 1, T, l, r, tgt, return_ntgt=False)
    P = 5
    ref = oracle_hgt(sd, 2, H, tgt[:P], nb[:P], codes, cen, A, b, n_store, l, r)["tgt"].numpy()
    assert np.abs(out["tgt"][:P] - ref).max() < 1e-4


def test_star_attn_kg_1024(dev):
    """k_g = 1024 stress (BASELINE configs[2]): the generic kernel path."""
    from gnnlm_amd import ops
    rs = np.random.RandomState(1)
    T, H, M, dsub, kg, N = 3, 8, 128, 8, 1024, 4000
    D = M * dsub
    codes = rs.randint(0, 256, size=(N, M)).astype(np.uint8)
    cen = (rs.randn(M, 256, dsub) * 0.5).astype(np.float32)
    U = (rs.randn(T, H, D) / np.sqrt(D)).astype(np.float32)
    ids = rs.randint(0, N, size=(T, kg)).astype(np.int64)
    ids[0, ::5] = -1
    Z, _ = ops.star_attn(torch.from_numpy(U).to(dev), torch.from_numpy(ids).to(dev),
                         codes=torch.from_numpy(codes).to(dev), centroids=torch.from_numpy(cen).to(dev))
    X = opq.pq_lookup(codes[np.where(ids < 0, 0, ids).reshape(-1)], cen).reshape(T, kg, D).astype(np.float64)
    s = np.where((ids >= 0)[:, None, :], np.einsum("tjd,thd->thj", X, U.astype(np.float64)), -np.inf)
    a = np.exp(s - s.max(-1, keepdims=True))
    a /= a.sum(-1, keepdims=True)
    assert np.abs(Z.cpu().numpy() - np.einsum("thj,tjd->thd", a, X)).max() < 2e-5


def test_knn_interp_degenerate(dev):
    from gnnlm_amd import ops
    vals = np.array([7, 8, 9], dtype=np.int32)
    lm = np.log(np.array([0.5, 0.25, 0.125], dtype=np.float32))
    sims = np.array([[0.3], [0.1], [0.9]], dtype=np.float32)
    ids = np.array([[-1], [1], [2]], dtype=np.int64)              # all-padding row, miss, hit
    tg = np.array([9, 7, 9], dtype=np.int64)                      # row 0: padding wraps to vals[-1] == 9 == target
    out, pk, rec = ops.knn_interp(*(torch.from_numpy(a).to(dev) for a in (lm, sims, ids, tg)), 0.01, 0.25,
                                  vals=torch.from_numpy(vals).to(dev))
    p_ref, r_ref = oknn.knn_target_prob(sims, ids, vals, tg, 0.01)
    ref = oknn.combine_knn_and_vocab_probs(p_ref, torch.from_numpy(lm), 0.25)
    assert np.array_equal(rec.cpu().numpy(), r_ref.numpy())
    np.testing.assert_allclose(pk.cpu().numpy(), p_ref.numpy(), rtol=1e-6)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-6, atol=1e-6)
