"""Parity at BASELINE.json's own sizes (GPU only).

(i)   the 103,227,021-row WikiText-103 store built on the device exactly as bench.py builds it (13.2 GB of codes,
      413 MB of labels): rows spread over the whole range -- including rows > 2^25 (byte offset >= 2^32), rows
      > 10^8 and the last row -- are pulled back to the host and `gnnlm_pq_gather_decode`, `gnnlm_star_attn`
      (PQ source) and `gnnlm_knn_interp` are checked against the oracle on exactly those rows (bit-exact for the
      byte / index work, <= 2e-5 for the float work);
(ii)  one full 256-token block (d = 1024, k_g = 128, l = r = 2, H = 8, PQ 128 x 8 + OPQ) on that store, ALL 256
      tokens, 1 and 3 HGT layers, against the un-elided oracle;
(iii) BASELINE configs[0] (N = 250,000 keys, k_g = 8, kNN k = 8, d = 1024, 1 layer) through `eval_lm.cli_main` on a
      data directory in the reference's on-disk formats;
(iv)  k_g = 1024 (configs[2]) through `gnnlm_hgt_forward`, not only the star kernel.
The reference's behaviour these follow: token_block_dataset.py:338-412, pq_wrapper.py:169-203, hgt.py:299-420,
knn_model.py:192-217, sequence_scorer.py:55-68."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import graph as og
from oracle import hgt as ohgt
from oracle import knn as oknn
from oracle import pq as opq
from oracle.hostrows import HostRows

N_FULL = 103_227_021
M, DSUB, D, H = 128, 8, 1024, 8
VOCAB = 267_744


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def full_store(dev):
    """bench.py's store: same generator, same seeds (bench.py::build)."""
    from gnnlm_amd.synthetic import device_codes, make_codec, zipf_dev
    codes = device_codes(N_FULL, M, dev, 1234)
    vgen = torch.Generator(device=dev)
    vgen.manual_seed(4321)
    vals = zipf_dev(N_FULL, VOCAB, vgen, dev).to(torch.int32)
    cen, A, b = make_codec(np.random.RandomState(1234), M, DSUB, D, opq=True)
    yield {"codes": codes, "vals": vals, "vals_host": vals.cpu().numpy(), "cen": cen, "A": A, "b": b}
    del codes, vals
    torch.cuda.empty_cache()


def spread_rows(rs, n):
    """n rows over the whole range; the fixed ones pin the 2^25-row (4 GiB byte offset) boundary, a row > 10^8
    and both ends."""
    special = [0, 1, 2, 2 ** 25 - 1, 2 ** 25, 2 ** 25 + 1, 2 ** 26 + 5, 100_000_001, N_FULL - 3, N_FULL - 2, N_FULL - 1]
    hi = rs.randint(2 ** 25, N_FULL, size=n // 2)
    lo = rs.randint(0, N_FULL, size=n - n // 2 - len(special))
    return np.concatenate([special, hi, lo]).astype(np.int64)


def test_full_store_gather_decode(dev, full_store):
    from gnnlm_amd import ops
    rs = np.random.RandomState(11)
    ids = spread_rows(rs, 10_000)
    ids = np.concatenate([ids, [-1, N_FULL, N_FULL + 5]])                  # not rows of the store: invalid slots
    rows, valid = og.slot_layout(ids.reshape(-1, 1), N_FULL, 2, 2)
    rows, valid = rows.reshape(-1), valid.reshape(-1)
    host = HostRows(full_store["codes"], rows[valid])
    out = ops.pq_gather_decode(full_store["codes"], torch.from_numpy(full_store["cen"]).to(dev),
                               torch.from_numpy(ids).to(dev), 2, 2, vals=full_store["vals"], want_codes=True,
                               want_labels=True)
    assert np.array_equal(out["valid"].cpu().numpy().astype(bool), valid)
    assert (rows[valid] >= 2 ** 25).sum() > 20_000                         # the test does reach past 4 GiB
    got_codes = out["codes"].cpu().numpy()
    assert np.array_equal(got_codes[valid], host[rows[valid]])             # bytes: exact
    x = out["x"].cpu().numpy()
    assert np.array_equal(x[valid], opq.pq_lookup(host[rows[valid]], full_store["cen"]))   # table look-up: exact
    assert not x[~valid].any()
    lab = out["labels"].cpu().numpy()
    assert np.array_equal(lab[valid], full_store["vals_host"][rows[valid]]) and (lab[~valid] == -1).all()


def star_reference(U, ids, X, ok):
    s = np.einsum("tjd,thd->thj", X, U.astype(np.float64))
    s = np.where(ok[:, None, :], s, -np.inf)
    with np.errstate(invalid="ignore"):
        a = np.exp(s - s.max(-1, keepdims=True))
        a = np.nan_to_num(a / a.sum(-1, keepdims=True))
    return np.einsum("thj,tjd->thd", a, X)


def test_full_store_star_attn(dev, full_store):
    from gnnlm_amd import ops
    rs = np.random.RandomState(12)
    T, kg = 96, 128
    ids = spread_rows(rs, T * kg).reshape(T, kg)
    ids[0, 5] = -1
    ids[1, :] = -1
    ids[2, 7] = N_FULL                                                     # >= n_store: not a neighbour (no OOB read)
    ids[2, 9] = N_FULL + 12345
    ok = (ids >= 0) & (ids < N_FULL)
    host = HostRows(full_store["codes"], ids[ok])
    U = (rs.randn(T, H, D) / np.sqrt(D)).astype(np.float32)
    Z, has = ops.star_attn(torch.from_numpy(U).to(dev), torch.from_numpy(ids).to(dev), codes=full_store["codes"],
                           centroids=torch.from_numpy(full_store["cen"]).to(dev))
    X = np.zeros((T, kg, D))
    X[ok] = opq.pq_lookup(host[ids[ok]], full_store["cen"])
    ref = star_reference(U, ids, X, ok)
    assert np.abs(Z.cpu().numpy() - ref).max() < 2e-5
    assert np.array_equal(has.cpu().numpy(), ok.any(1).astype(np.float32))


def test_full_store_mapped_shards(dev, full_store):
    """The 103 M-row table as EIGHT mapped shards (views of the one tensor with two halo rows each side, i.e. the pointer
    table dist.PeerMappedFetcher builds from its peers' memory): star attention and the slot decode through the shard
    table reproduce the one-table results bit for bit -- shard bases beyond 4 GiB, rows at both ends of every shard."""
    from gnnlm_amd import _lib, ops
    from gnnlm_amd.dist import Shard
    from gnnlm_amd.hgt import CodeStore, shards_device_ptr
    rs = np.random.RandomState(14)
    W = 8
    shs = [Shard(N_FULL, W, g, halo_left=2, halo_right=2) for g in range(W)]
    views = [(full_store["codes"][s_.store_row0:s_.store_row0 + s_.store_rows], s_.store_row0) for s_ in shs]
    edges = np.array([r for s_ in shs for r in (s_.row0 - 1, s_.row0, s_.row0 + 1, s_.row0 + s_.n_local - 1) if 0 <= r < N_FULL])
    T, kg = 64, 128
    ids = np.concatenate([edges, spread_rows(rs, T * kg - len(edges))]).astype(np.int64).reshape(T, kg)
    ids[0, 5], ids[1, :], ids[2, 7] = -1, -1, N_FULL + 3
    cen = torch.from_numpy(full_store["cen"]).to(dev)
    U = torch.from_numpy((rs.randn(T, H, D) / np.sqrt(D)).astype(np.float32)).to(dev)
    idt = torch.from_numpy(ids).to(dev)
    Z0, h0 = ops.star_attn(U, idt, codes=full_store["codes"], centroids=cen)
    Z1, h1 = ops.star_attn(U, idt, centroids=cen, n_store=N_FULL, shards=views, rows_per_rank=shs[0].per)
    assert torch.equal(Z0, Z1) and torch.equal(h0, h1)
    # slot decode (the L > 1 path) through the shard table, straight on the C ABI
    st = CodeStore(codes=None, centroids=cen, n_store=N_FULL, shards=views, rows_per_rank=shs[0].per)
    cids = torch.from_numpy(ids.reshape(-1)[:4096].copy()).to(dev)
    ref = ops.pq_gather_decode(full_store["codes"], cen, cids, 2, 2, want_codes=True)
    g = _lib.gnnlm_gather_t()
    n_slots = cids.numel() * 5
    x = torch.empty(n_slots, D, device=dev)
    oc = torch.empty(n_slots, M, device=dev, dtype=torch.uint8)
    ov = torch.empty(n_slots, device=dev, dtype=torch.uint8)
    g.shards, g.n_store, g.M, g.dsub, g.centroids = shards_device_ptr(st), N_FULL, M, DSUB, cen.data_ptr()
    g.ids, g.n_groups, g.left, g.right, g.vals_itemsize = cids.data_ptr(), cids.numel(), 2, 2, 4
    g.out_x, g.ld_x, g.out_codes, g.out_valid = x.data_ptr(), D, oc.data_ptr(), ov.data_ptr()
    _lib.call_desc("gnnlm_pq_gather_decode", g)
    assert torch.equal(x, ref["x"]) and torch.equal(oc, ref["codes"]) and torch.equal(ov, ref["valid"])


def test_full_store_knn_interp(dev, full_store):
    from gnnlm_amd import ops
    rs = np.random.RandomState(13)
    n, k = 256, 1024
    ids = spread_rows(rs, n * k).reshape(n, k)
    ids[:, :4] = rs.randint(10 ** 8, N_FULL, size=(n, 4))                  # labels at rows > 10^8 in every row
    ids[::5, -3:] = -1                                                     # padding: wraps to the LAST row like numpy
    vals = full_store["vals_host"]
    sims = np.sort(rs.uniform(0.2, 0.9, size=(n, k)).astype(np.float32), axis=1)[:, ::-1].copy()
    targets = np.where(rs.rand(n) < 0.5, vals[ids[:, 2]], rs.randint(0, VOCAB, size=n)).astype(np.int64)
    lm = np.log(rs.uniform(1e-4, 1, size=n)).astype(np.float32)
    for t, lmbda in [(0.01, 0.1), (1.0, 0.25)]:
        p_ref, rec_ref = oknn.knn_target_prob(sims, ids, vals, targets, t)
        ref = oknn.combine_knn_and_vocab_probs(p_ref, torch.from_numpy(lm), lmbda)
        out, pk, rec = ops.knn_interp(*(torch.from_numpy(a).to(dev) for a in (lm, sims, ids, targets)), t, lmbda,
                                      vals=full_store["vals"])
        assert np.array_equal(rec.cpu().numpy(), rec_ref.numpy())
        np.testing.assert_allclose(pk.cpu().numpy(), p_ref.numpy(), rtol=5e-5, atol=1e-7)
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=2e-5, atol=5e-6)


def test_knn_interp_rows_outside_store(dev):
    """ids >= n_store (the reference raises IndexError, knn_model.py:198) are not read: they never match."""
    from gnnlm_amd import ops
    vals = np.array([7, 8, 9], dtype=np.int32)
    lm = np.log(np.array([0.5, 0.25], dtype=np.float32))
    sims = np.array([[0.3, 0.2], [0.1, 0.05]], dtype=np.float32)
    ids = np.array([[3, 1], [2, 10 ** 12]], dtype=np.int64)
    tg = np.array([8, 9], dtype=np.int64)
    out, pk, rec = ops.knn_interp(*(torch.from_numpy(a).to(dev) for a in (lm, sims, ids, tg)), 1.0, 0.25,
                                  vals=torch.from_numpy(vals).to(dev))
    e = np.exp(sims - sims.max(1, keepdims=True))
    p = e / e.sum(1, keepdims=True)
    np.testing.assert_allclose(pk.cpu().numpy(), [p[0, 1], p[1, 0]], rtol=1e-6)
    assert rec.cpu().tolist() == [1, 1]


def oracle_block(sd, L, tgt, nb, host_codes, cen, A, b, dtype):
    gr = og.build_graph(nb, np.zeros(nb.shape[0], np.int64), N_FULL, 2, 2)
    ntgt = opq.pq_lookup(host_codes[gr["ntgt_offsets"]], cen).astype(np.float64)
    ntgt = (ntgt - b.astype(np.float64)) @ A.astype(np.float64)
    feats = {"tgt": torch.from_numpy(tgt.astype(np.float64)).to(dtype), "ntgt": torch.from_numpy(ntgt).to(dtype)}
    return ohgt.hgt_forward({k: torch.as_tensor(v).to(dtype) for k, v in sd.items()}, L, H, feats, gr)["tgt"]


@pytest.mark.parametrize("L,dtype,tol", [(1, torch.float64, 1e-4), (3, torch.float64, 2e-4)])
def test_full_block_all_tokens(dev, full_store, L, dtype, tol):
    """All 256 tokens of a WikiText-103 block on the full store against the un-elided float64 oracle (163,840 ntgt
    nodes; ~25 GB of host memory and ~1 min on the GPU box's 256 cores at L = 3)."""
    from tests.test_hgt_gpu import make_store, run_hip
    rs = np.random.RandomState(1234 + L)
    T, kg = 256, 128
    nb = rs.randint(0, N_FULL, size=(T, kg)).astype(np.int64)
    nb[rs.rand(T, kg) < 0.001] = -1
    nb[3] = -1
    nb[0, :6] = [0, 1, N_FULL - 1, N_FULL - 2, 2 ** 25, 100_000_001]       # clipped contexts at both ends, > 4 GiB offsets
    tgt = rs.randn(T, D).astype(np.float16).astype(np.float32)
    rows, valid = og.slot_layout(nb, N_FULL, 2, 2)
    host = HostRows(full_store["codes"], rows[valid])
    from gnnlm_amd.hgt import CodeStore
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    store = CodeStore(codes=full_store["codes"], centroids=t(full_store["cen"]), n_store=N_FULL, vals=full_store["vals"],
                      A=t(full_store["A"]), b=t(full_store["b"]))
    sd = {k: v.numpy() for k, v in ohgt.init_hgt_weights(L, D, H, seed=7).items()}
    out = run_hip(dev, sd, L, H, D, store, nb, 1, T, 2, 2, tgt, return_ntgt=False)["tgt"]
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 128)))
    ref = oracle_block(sd, L, tgt, nb, host, full_store["cen"], full_store["A"], full_store["b"], dtype).double().numpy()
    err = np.abs(out - ref).max()
    assert np.isfinite(out).all() and err < tol, (L, err)


def test_config0_plumbing_shape_through_eval_lm(dev, tmp_path):
    """BASELINE configs[0]: valid-only datastore of 250,000 keys, k_g = 8, kNN k = 8, d = 1024, 1 HGT layer,
    `eval_lm --graph --use-precompute-feat --knnlm` end to end against the oracle."""
    from gnnlm_amd import eval_lm
    from gnnlm_amd.synthetic import make_problem
    from oracle import pipeline
    from tests.test_mirrors_gpu import write_dstore
    n_train, n_test, kg, k, T, L = 250_000, 300, 8, 8, 256, 1
    V, cutoff = 30_000, [2000, 8000]
    prob = make_problem(n_store=n_train, d=D, n_heads=H, M=M, dsub=DSUB, vocab=V, cutoff=cutoff, T=n_test, kg=kg, left=2,
                        right=2, n_layers=L, k=k, seed=5)
    data = tmp_path / "data-bin"
    rs = np.random.RandomState(0)
    train_keys = rs.randn(n_train, 64).astype(np.float16)                  # kNN keys: only their labels matter below
    write_dstore(str(data / "train_dstore"), train_keys, prob["vals"].astype(np.int16), V)
    np.save(str(data / "train_dstore" / "quantized-keys.npy"), prob["codes"])
    blk = prob["block"]
    blk["targets"] = np.maximum(blk["targets"], 4)
    write_dstore(str(data / "test_dstore"), blk["tgt_feats"], blk["targets"].astype(np.int16), V)
    blk["ids"].tofile(str(data / "test_dstore" / f"neighbors.mmap.{kg}"))
    sd = {"decoder.hgt_decoder." + k_: v for k_, v in prob["sd"].items()}
    w = prob["asm"]
    for i, e in enumerate(w["emb"]):
        sd[f"decoder.embed_tokens.embeddings.{i}.0.weight"] = e
        if i:
            sd[f"decoder.embed_tokens.embeddings.{i}.1.weight"] = w["proj"][i]
    sd["decoder.adaptive_softmax.head.class_proj.weight"] = w["class_proj"]
    sd["decoder.tgt_quantizer.centroids_torch"] = torch.from_numpy(prob["cen"])
    sd["decoder.tgt_quantizer.A"] = torch.from_numpy(prob["A"])
    sd["decoder.tgt_quantizer.b"] = torch.from_numpy(prob["b"])
    margs = Namespace(decoder_embed_dim=D, decoder_attention_heads=H, graph_layer=L, decoder_gcn_dim=D,
                      adaptive_softmax_cutoff=",".join(map(str, cutoff)), orig_prob_ratio=0.0, short_cut=False,
                      quantizer_path="")
    torch.save({"args": margs, "model": sd}, str(tmp_path / "ckpt.pt"))
    model = {"sd": prob["sd"], "n_layers": L, "n_heads": H, "centroids": prob["cen"], "A": prob["A"], "b": prob["b"],
             "codes": prob["codes"], "vals": prob["vals"], "n_store": n_train, "left": 2, "right": 2, "asm": w}
    lam, temp = 0.25, 1.0
    ref = []
    for s in range(0, n_test, T):
        e = min(n_test, s + T)
        one = {"neighbor_idxs": blk["ids"][s:e], "tgt_feats": blk["tgt_feats"][s:e], "targets": blk["targets"][s:e],
               "knn_sims": blk["knn_sims"][s:e], "knn_ids": blk["knn_ids"][s:e]}
        ref.append(pipeline.eval_block(one, model, lam, temp)["logp"])
    ref = torch.cat(ref).double()

    class Replay:                                   # the search results of the synthetic problem (faiss contract)
        def __init__(self):
            self.pos = 0

        def search(self, q, kk):
            n = q.shape[0]
            sl = slice(self.pos, self.pos + n)
            self.pos += n
            return blk["knn_sims"][sl, :kk].copy(), blk["knn_ids"][sl, :kk].copy()

    from gnnlm_amd.knn_model import KNNModel
    knn = KNNModel("faiss_store.ip", str(data / "train_dstore"), k=k, metric_type="do_not_recomp_ip", index=Replay(),
                   no_load_keys=True, device=dev)
    args = eval_lm.get_parser().parse_args(
        [str(data), "--path", str(tmp_path / "ckpt.pt"), "--gen-subset", "test", "--graph", "--neighbor-context", "2",
         "--gcn-k", str(kg), "--use-precompute-feat", "--sample-break-mode", "none", "--max-tokens", str(T),
         "--tokens-per-sample", str(T), "--gcn-context-window", "0", "--knn-keytype", "gcn_feat", "--model-overrides",
         "{'orig_prob_ratio': 0.0}", "--knnlm", "--k", str(k), "--lmbda", str(lam), "--dstore-dir",
         str(data / "train_dstore"), "--temperature", str(temp), "--knn-sim-func", "do_not_recomp_ip"])
    args.knn_model = knn
    res = eval_lm.main(args)
    assert res["count"] == n_test
    assert abs(res["score_sum"] - ref.sum().item()) < 1e-4 * n_test
    assert abs(res["ppl"] - 2 ** (-ref.sum().item() / n_test / np.log(2))) < 0.02


@pytest.mark.parametrize("L", [1, 2])
def test_hgt_forward_kg_1024(dev, L):
    """configs[2]'s k_g = 1024 through gnnlm_hgt_forward (star attention over 1024 neighbours per token, the
    context expansion of 1024 x 5 slots per token at L = 2) against the float64 oracle."""
    from tests.test_hgt_gpu import make_store, oracle_hgt, run_hip
    d, Hh, Mm, dsub, T, kg, l, r = 256, 8, 32, 8, 6, 1024, 2, 2
    rs = np.random.RandomState(77 + L)
    n_store = 50_000
    codes = rs.randint(0, 256, size=(n_store, Mm)).astype(np.uint8)
    cen = (rs.randn(Mm, 256, dsub) * 0.5).astype(np.float32)
    A = (rs.randn(Mm * dsub, d) / np.sqrt(Mm * dsub)).astype(np.float32)
    b = (rs.randn(Mm * dsub) * 0.1).astype(np.float32)
    nb = rs.randint(0, n_store, size=(T, kg)).astype(np.int64)
    nb[rs.rand(T, kg) < 0.01] = -1
    nb[2] = -1
    tgt = rs.randn(T, d).astype(np.float16).astype(np.float32)
    sd = {k: v.numpy() for k, v in ohgt.init_hgt_weights(L, d, Hh, seed=3).items()}
    out = run_hip(dev, sd, L, Hh, d, make_store(dev, codes, cen, A, b), nb, 1, T, l, r, tgt, return_ntgt=False)
    ref = oracle_hgt(sd, L, Hh, tgt, nb, codes, cen, A, b, n_store, l, r)["tgt"].numpy()
    assert np.abs(out["tgt"] - ref).max() < 1e-4
