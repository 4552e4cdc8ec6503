"""Host-side mirrors on the GPU: TorchPQCodec.decode, KNNModel.get_knn_prob, SequenceScorer.generate and
the eval_lm driver, against the reference-generated golden vectors and the CPU oracle."""
import json
import os
import types
from argparse import Namespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import knn as oknn


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("case", ["sq_pre", "sq_pre_nob", "rect_pre", "nopre"])
def test_codec_decode_golden(dev, golden, case):
    from gnnlm_amd.pq_wrapper import TorchPQCodec
    g = golden("pq")
    A = g[f"{case}.A"] if f"{case}.A" in g else None
    b = g[f"{case}.b"] if f"{case}.b" in g else None
    c = TorchPQCodec.from_arrays(g[f"{case}.cen"], A, b).to(dev)
    out = c.decode(torch.from_numpy(g[f"{case}.codes"]).to(dev)).cpu().numpy()
    np.testing.assert_allclose(out, g[f"{case}.decode"], atol=2e-6, rtol=1e-5)            # pq_wrapper.py:233-237 uses 1e-6


@pytest.mark.parametrize("case", ["sq_pre", "sq_pre_nob", "rect_pre", "nopre"])
def test_codec_encode_hip(dev, golden, case):
    """HIP encode == the reference's codes (integer output: exact), and == the oracle on a larger batch
    wherever the two best distances are not within float rounding of each other."""
    from gnnlm_amd.pq_wrapper import TorchPQCodec
    from oracle import pq as opq_
    g = golden("pq")
    A = g[f"{case}.A"] if f"{case}.A" in g else None
    b = g[f"{case}.b"] if f"{case}.b" in g else None
    c = TorchPQCodec.from_arrays(g[f"{case}.cen"], A, b).to(dev)
    codes = c.encode(torch.from_numpy(g[f"{case}.x"].copy()).to(dev)).cpu().numpy()
    assert np.array_equal(codes, g[f"{case}.codes"])
    rs = np.random.RandomState(0)
    x = rs.randn(2000, g[f"{case}.x"].shape[1]).astype(np.float32)
    got = c.encode(torch.from_numpy(x).to(dev)).cpu().numpy()
    ref = opq_.pq_encode(x, g[f"{case}.cen"], A, b)
    assert (got != ref).mean() < 1e-3                      # near-ties may resolve differently in float32
    dec_got = opq_.pq_lookup(got, g[f"{case}.cen"]); dec_ref = opq_.pq_lookup(ref, g[f"{case}.cen"])
    xr = x @ A.T + (b if b is not None and b.size else 0) if A is not None else x
    assert np.allclose(((xr - dec_got) ** 2).sum(1), ((xr - dec_ref) ** 2).sum(1), rtol=1e-4, atol=1e-4)   # equally good codes


def test_codec_decode_full_size(dev, golden):
    from gnnlm_amd.pq_wrapper import TorchPQCodec
    from tests.test_oracle_golden import full_size_codec
    g = golden("pq")
    cen, A, b = full_size_codec()
    c = TorchPQCodec.from_arrays(cen, A, b).to(dev)
    out = c.decode(torch.from_numpy(g["full.codes"]).to(dev)).cpu().numpy()
    np.testing.assert_allclose(out, g["full.decode"], atol=3e-5, rtol=1e-5)


class FixedIndex:
    """Replays recorded search results (faiss contract)."""

    def __init__(self, dists, ids):
        self.d, self.i = dists, ids

    def search(self, q, k):
        return self.d[:, :k].copy(), self.i[:, :k].copy()


def write_dstore(path, keys, vals, vocab, fp16=True):
    os.makedirs(path, exist_ok=True)
    keys.tofile(os.path.join(path, "keys.npy"))
    vals.tofile(os.path.join(path, "vals.npy"))
    json.dump({"dstore_size": len(vals), "hidden_size": keys.shape[1], "vocab_size": vocab, "dstore_fp16": fp16,
               "val_size": 1}, open(os.path.join(path, "info.json"), "w"))


@pytest.mark.parametrize("metric_type", ["do_not_recomp_ip", "do_not_recomp_l2", "ip", "l2"])
def test_knn_model_golden(dev, golden, tmp_path, metric_type):
    from gnnlm_amd.knn_model import KNNModel
    g = golden("knn")
    write_dstore(str(tmp_path / "d"), g["keys"], g["vals"].astype(np.int16), 50)
    for cosine in (False, True):
        for t in (1.0, 0.01):
            tag = f"{metric_type}.{'cos' if cosine else 'raw'}.t{t}"
            m = KNNModel("faiss_store.cosine" if cosine else "faiss_store.ip", str(tmp_path / "d"), k=8,
                         metric_type=metric_type, index=FixedIndex(g[tag + ".dists"], g[tag + ".ids"]), device=dev)
            p, rec = m.get_knn_prob(torch.from_numpy(g["queries"]).to(dev), targets=torch.from_numpy(g["targets"]).to(dev),
                                    t=t, return_recall=True)
            np.testing.assert_allclose(p.cpu().numpy(), g[tag + ".p"], rtol=3e-5, atol=1e-7)
            assert np.array_equal(rec.cpu().numpy(), g[tag + ".recall"])
            dense, sims, knns = m.get_knn_prob(torch.from_numpy(g["queries"]).to(dev), t=t, return_knn=True)
            np.testing.assert_allclose(dense.cpu().numpy(), g[tag + ".dense"], rtol=1e-3, atol=2e-5)  # t=0.01 amplifies 100x
    with pytest.raises(ValueError):
        KNNModel("x", str(tmp_path / "nope"), index=FixedIndex(None, None))


def test_exact_index_matches_oracle_search(dev, golden):
    from gnnlm_amd.knn_model import ExactIndex
    g = golden("knn")
    for metric, cosine in [("ip", False), ("ip", True), ("l2", False)]:
        q = oknn.normalize_queries(torch.from_numpy(g["queries"]), cosine).numpy()
        d_ref, i_ref = oknn.brute_force_search(q, g["keys"], 8, metric, cosine)
        d, i = ExactIndex(g["keys"], metric, cosine, dev).search(q, 8)
        assert np.array_equal(i, i_ref)
        np.testing.assert_allclose(d, d_ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("keytype", ["gcn_feat", "keytype"])
@pytest.mark.parametrize("lmbda,temp", [(0.25, 1.0), (0.1, 0.01), (0.0, 1.0)])
def test_sequence_scorer_golden(dev, golden, tmp_path, keytype, lmbda, temp):
    """SequenceScorer mirror vs the reference's own SequenceScorer.generate (scripted model)."""
    from gnnlm_amd.adaptive_softmax import AdaptiveSoftmax
    from gnnlm_amd.knn_model import ExactIndex, KNNModel
    from gnnlm_amd.model import GnnLmModel
    from gnnlm_amd.sequence_scorer import SequenceScorer
    g, ga = golden("scorer"), golden("adaptive_softmax")
    emb = [torch.from_numpy(ga[f"emb{i}"]) for i in range(3)]
    proj = [None] + [torch.from_numpy(ga[f"proj{i}"]) for i in (1, 2)]
    asm = AdaptiveSoftmax(list(ga["cutoff"]), emb, proj, torch.from_numpy(ga["class_proj"]), dev)
    feats, inner = torch.from_numpy(g["feats"]).to(dev), torch.from_numpy(g["inner"]).to(dev)

    class Scripted(GnnLmModel):
        def __init__(self):
            torch.nn.Module.__init__(self)
            self.adaptive_softmax = asm

        def forward(self, src_tokens, src_lengths=None, graph=None):
            return feats, {"inner_states": [inner], "gcn_feat": feats.transpose(0, 1)}

    write_dstore(str(tmp_path / "d"), g["keys"], g["vals"].astype(np.int16), 24)

    class PaddedExact(ExactIndex):               # the golden's stand-in pads every third row with one -1
        def search_device(self, q, k):
            d, i = super().search_device(q, k)
            i[::3, -1:] = -1
            d[::3, -1:] = -3.4e38
            return d, i

    knn = KNNModel("faiss_store.cosine", str(tmp_path / "d"), k=6, metric_type="do_not_recomp_ip",
                   index=PaddedExact(g["keys"], "ip", True, dev), device=dev)
    target = torch.from_numpy(g["target"]).to(dev)
    bsz, T = target.shape
    sample = {"id": torch.arange(bsz), "nsentences": bsz, "ntokens": bsz * T,
              "net_input": {"src_tokens": torch.zeros(bsz, T, dtype=torch.long, device=dev),
                            "src_lengths": torch.full((bsz,), T)},
              "target": target, "start_indices": torch.tensor([[0], [2]])}
    tgt_dict = types.SimpleNamespace(pad=lambda: 1, eos=lambda: 2)
    scorer = SequenceScorer(tgt_dict, softmax_batch=3072, args=Namespace(lmbda=lmbda, knn_keytype=keytype))
    hyp = scorer.generate([Scripted()], sample, knn_dstore=knn, temperature=temp)
    for i, h in enumerate(hyp):
        h = h[0]
        tag = f"{keytype}.l{lmbda}.t{temp}.{i}"
        assert np.array_equal(h["tokens"].cpu().numpy(), g[tag + ".tokens"])
        np.testing.assert_allclose(h["positional_scores"].cpu().numpy(), g[tag + ".positional_scores"], rtol=3e-5, atol=5e-6)
        np.testing.assert_allclose(h["score"].item(), g[tag + ".score"], rtol=3e-5, atol=5e-6)
        np.testing.assert_allclose(h["dstore_keys"].cpu().numpy(), g[tag + ".dstore_keys"])
        if lmbda > 0:
            assert np.array_equal(h["knn_recall"].cpu().numpy(), g[tag + ".knn_recall"])
        else:
            assert h["knn_recall"] is None


def make_data_dir(tmp_path, n_test=41, L=2):
    """A synthetic data directory in the reference's on-disk formats + a reference-style checkpoint."""
    from gnnlm_amd.synthetic import make_problem
    d, H, M, dsub, V, kg, T = 64, 4, 16, 4, 600, 6, 16
    n_train = 2000                                  # n_test = 41: 2 full blocks + a 9-token block
    prob = make_problem(n_store=n_train, d=d, n_heads=H, M=M, dsub=dsub, vocab=V, cutoff=[100, 300], T=n_test, kg=kg,
                        left=2, right=2, n_layers=L, k=8, seed=3)
    data = tmp_path / "data-bin"
    rs = np.random.RandomState(0)
    train_keys = rs.randn(n_train, d).astype(np.float16)
    write_dstore(str(data / "train_dstore"), train_keys, prob["vals"].astype(np.int16), V)   # fp16 store, V < 2**15 -> int16
    np.save(str(data / "train_dstore" / "quantized-keys.npy"), prob["codes"])
    blk = prob["block"]
    blk["targets"] = np.maximum(blk["targets"], 4)        # ids 0-3 are fairseq's specials; pad (1) is stripped by the scorer
    write_dstore(str(data / "test_dstore"), blk["tgt_feats"], blk["targets"].astype(np.int16), V)
    blk["ids"].tofile(str(data / "test_dstore" / f"neighbors.mmap.{kg}"))
    sd = {"decoder.hgt_decoder." + k: v for k, v in prob["sd"].items()}
    w = prob["asm"]
    for i, e in enumerate(w["emb"]):
        sd[f"decoder.embed_tokens.embeddings.{i}.0.weight"] = e
        if i:
            sd[f"decoder.embed_tokens.embeddings.{i}.1.weight"] = w["proj"][i]
    sd["decoder.adaptive_softmax.head.class_proj.weight"] = w["class_proj"]
    sd["decoder.tgt_quantizer.centroids_torch"] = torch.from_numpy(prob["cen"])
    sd["decoder.tgt_quantizer.A"] = torch.from_numpy(prob["A"])
    sd["decoder.tgt_quantizer.b"] = torch.from_numpy(prob["b"])
    margs = Namespace(decoder_embed_dim=d, decoder_attention_heads=H, graph_layer=L, decoder_gcn_dim=d,
                      adaptive_softmax_cutoff="100,300", orig_prob_ratio=0.0, short_cut=False, quantizer_path="")
    torch.save({"args": margs, "model": sd}, str(tmp_path / "ckpt.pt"))
    base = [str(data), "--path", str(tmp_path / "ckpt.pt"), "--gen-subset", "test", "--graph", "--neighbor-context", "2",
            "--gcn-k", str(kg), "--use-precompute-feat", "--sample-break-mode", "none", "--max-tokens", str(2 * T),
            "--tokens-per-sample", str(T), "--gcn-context-window", "0", "--knn-keytype", "gcn_feat",
            "--model-overrides", "{'orig_prob_ratio': 0.0}"]
    model = {"sd": prob["sd"], "n_layers": L, "n_heads": H, "centroids": prob["cen"], "A": prob["A"], "b": prob["b"],
             "codes": prob["codes"], "vals": prob["vals"], "n_store": n_train, "left": 2, "right": 2, "asm": w}
    return dict(prob=prob, blk=blk, data=data, base=base, model=model, train_keys=train_keys, T=T, n_test=n_test,
                n_train=n_train, w=w, L=L, H=H)


def test_eval_lm_gcn_context_window(dev, tmp_path):
    """--gcn-context-window w: every block is prefixed by the w previous tokens, which take part in the
    graph but are not scored (token_block_dataset.py:264-285,331; sequence_scorer.py:156-162)."""
    from gnnlm_amd import eval_lm
    from oracle import pipeline
    c = make_data_dir(tmp_path)
    blk, T, n_test, w_ctx = c["blk"], c["T"], c["n_test"], 5
    total = 0.0
    for s in range(0, n_test, T):
        e = min(n_test, s + T)
        cs = max(0, s - w_ctx) if s > 0 else s
        one = {"neighbor_idxs": blk["ids"][cs:e], "tgt_feats": blk["tgt_feats"][cs:e], "targets": blk["targets"][cs:e],
               "knn_sims": None, "knn_ids": None}
        total += pipeline.eval_block(one, c["model"], 0.0, 1.0)["lm_logp"][s - cs:].double().sum().item()
    base = list(c["base"])
    base[base.index("--gcn-context-window") + 1] = str(w_ctx)
    base[base.index("--max-tokens") + 1] = str(T + w_ctx)
    res = eval_lm.cli_main(base)
    assert res["count"] == n_test
    assert abs(res["score_sum"] - total) < 1e-4 * n_test


def test_eval_lm_end_to_end(dev, tmp_path):
    """A synthetic data directory in the reference's on-disk formats -> the driver's ppl equals the
    oracle's (float32 accumulation differences aside), with and without kNN, incl. a ragged last block."""
    from gnnlm_amd import eval_lm
    from oracle import knn as oknn_, pipeline
    c = make_data_dir(tmp_path)
    prob, blk, data, base, model, train_keys = c["prob"], c["blk"], c["data"], c["base"], c["model"], c["train_keys"]
    T, n_test, L, H, w = c["T"], c["n_test"], c["L"], c["H"], c["w"]
    # oracle, block by block
    lam, temp, k = 0.25, 1.0, 8
    ref_lm, ref_mix = [], []
    for s in range(0, n_test, T):
        e = min(n_test, s + T)
        one = {"neighbor_idxs": blk["ids"][s:e], "tgt_feats": blk["tgt_feats"][s:e], "targets": blk["targets"][s:e]}
        o = pipeline.eval_block(dict(one, knn_sims=None, knn_ids=None), model, 0.0, 1.0)
        ref_lm.append(o["lm_logp"])
        q = oknn_.normalize_queries(o["gcn_feat"].float(), True).numpy()
        dd, ii = oknn_.brute_force_search(q, train_keys, k, "ip", cosine=True)
        p, _ = oknn_.knn_target_prob(dd, ii, prob["vals"], blk["targets"][s:e], temp)
        ref_mix.append(oknn_.combine_knn_and_vocab_probs(p, o["lm_logp"], lam))
    ref_lm, ref_mix = torch.cat(ref_lm).double(), torch.cat(ref_mix).double()
    res = eval_lm.cli_main(base)
    assert res["count"] == n_test
    assert abs(res["score_sum"] - ref_lm.sum().item()) < 1e-4 * n_test
    assert abs(res["ppl"] - 2 ** (-ref_lm.sum().item() / n_test / np.log(2))) < 0.02      # north_star: ppl within 0.02
    # one block per batch as in the recipe (--max-tokens 256 == --tokens-per-sample 256): with bsz > 1 the
    # scorer reproduces the reference's [T,B]-vs-[B,T] target order quirk (sequence_scorer.py:117)
    base1 = [a for a in base]
    base1[base1.index("--max-tokens") + 1] = str(T)
    res = eval_lm.cli_main(base1 + ["--knnlm", "--k", str(k), "--lmbda", str(lam), "--dstore-dir", str(data / "train_dstore"),
                                   "--index-file", str(data / "train_dstore" / "faiss_store.cosine"),
                                   "--temperature", str(temp), "--knn-sim-func", "ip"])
    assert abs(res["score_sum"] - ref_mix.sum().item()) < 2e-4 * n_test
    # --batch-blocks (this build): several of those one-block batches per launch, same scores (incl. the kNN pairing)
    knn_args = ["--knnlm", "--k", str(k), "--lmbda", str(lam), "--dstore-dir", str(data / "train_dstore"),
                "--index-file", str(data / "train_dstore" / "faiss_store.cosine"), "--temperature", str(temp), "--knn-sim-func", "ip"]
    # (default: one-block batches are scored 32 per launch -- `res` above; 0 switches that off, N sets the group size)
    for nb_ in ("0", "4"):
        res4 = eval_lm.cli_main(base1 + knn_args + ["--batch-blocks", nb_])
        assert res4["count"] == n_test and abs(res4["score_sum"] - res["score_sum"]) < 1e-6 * n_test
    # the recipe's own --softmax-batch (3072 for its 256-token batches, hgt_lm_wiki103_reproduce.sh:84): the reference's precondition
    # B * T < softmax_batch is about the recipe's batch, not about how many of them this build scores per launch
    res_sb = eval_lm.cli_main(base1 + knn_args + ["--softmax-batch", str(T + 1)])
    assert res_sb["count"] == n_test and abs(res_sb["score_sum"] - res["score_sum"]) < 1e-6 * n_test
    # --graph-capture (this build): forward and softmax of the recipe's literal one-block batches replayed from HIP graphs, one
    # pair per batch shape (the 16-token blocks and the ragged 9-token one) -- the same scores to the bit, run after run
    plain = eval_lm.cli_main(base1 + knn_args + ["--batch-blocks", "0"])
    for extra in ([], ["--streams", "1"], ["--streams", "3"]):          # (default for one-block batches: 6 streams in turn)
        cap = eval_lm.cli_main(base1 + knn_args + ["--batch-blocks", "0", "--graph-capture"] + extra)
        assert cap["count"] == n_test and abs(cap["score_sum"] - plain["score_sum"]) <= 1e-12 * abs(plain["score_sum"])
    three = eval_lm.cli_main(base1 + knn_args + ["--batch-blocks", "0", "--streams", "3"])       # streams without graphs
    assert abs(three["score_sum"] - plain["score_sum"]) <= 1e-12 * abs(plain["score_sum"])
    # --num-shards / --shard-id partition the blocks (eval_lm.py:131-132)
    parts = [eval_lm.cli_main(base + ["--num-shards", "2", "--shard-id", str(i)]) for i in range(2)]
    assert sum(p["count"] for p in parts) == n_test
    assert abs(sum(p["score_sum"] for p in parts) - ref_lm.sum().item()) < 1e-4 * n_test


def test_find_knn_producer(dev, tmp_path):
    """neighbors.mmap.{k} written by the find_knn mirror == exact search of the oracle (cosine index,
    un-normalised queries as in knn/find_knn.py:63-65), incl. the truncated copies."""
    from gnnlm_amd import find_knn
    rs = np.random.RandomState(4)
    d, n_train, n_test, k = 32, 700, 90, 16
    train = rs.randn(n_train, d).astype(np.float16)
    test = rs.randn(n_test, d).astype(np.float16)
    data = tmp_path / "data-bin"
    write_dstore(str(data / "train_dstore"), train, rs.randint(4, 100, n_train).astype(np.int16), 100)
    write_dstore(str(data / "test_dstore"), test, rs.randint(4, 100, n_test).astype(np.int16), 100)
    f = find_knn.main(find_knn.get_parser().parse_args(["--data-dir", str(data), "--subset", "test", "--k", str(k),
                                                        "--bsz", "37", "--truncate-to", "8", "4"]))
    got = np.memmap(f, dtype=np.int64, mode="r", shape=(n_test, k))
    _, ref = oknn.brute_force_search(test.astype(np.float32), train, k, "ip", cosine=True)
    assert np.array_equal(np.array(got), ref)
    for k2 in (8, 4):
        t = np.memmap(str(data / "test_dstore" / f"neighbors.mmap.{k2}"), dtype=np.int64, mode="r", shape=(n_test, k2))
        assert np.array_equal(np.array(t), ref[:, :k2])


@pytest.mark.parametrize("fp16", [True, False])
def test_eval_lm_save_knnlm_dstore(dev, tmp_path, fp16):
    """--save-knnlm-dstore (fairseq_cli/eval_lm.py:103-104,178-205,222-242,322-323) on the precomputed-feature path: the
    split's kNN datastore of the GNN features -- info.json, keys.npy ([n, d], the `--knn-keytype gcn_feat` rows the
    scorer hands back, incl. the ragged last block), vals.npy ([n, 1], the target tokens, int16 for an fp16 store with a
    small vocabulary) -- equals what the oracle computes, and the DataStore mirror reads it back."""
    import json
    from gnnlm_amd import eval_lm
    from gnnlm_amd.data_store import DataStore
    from oracle import pipeline
    c = make_data_dir(tmp_path)
    blk, T, n_test = c["blk"], c["T"], c["n_test"]
    out = tmp_path / "dstores"
    res = eval_lm.cli_main(c["base"] + ["--save-knnlm-dstore", "--dstore-mmap", str(out)] + (["--dstore-fp16"] if fp16 else []))
    assert res["count"] == n_test
    sdir = out / "test_dstore-gcn_feat"
    info = json.load(open(sdir / "info.json"))
    d = c["prob"]["d"]
    assert info == {"dstore_size": n_test, "hidden_size": d, "vocab_size": 600, "dstore_fp16": fp16, "val_size": 1}
    keys = np.memmap(sdir / "keys.npy", dtype=np.float16 if fp16 else np.float32, mode="r", shape=(n_test, d))
    vals = np.memmap(sdir / "vals.npy", dtype=np.int16 if fp16 else np.int32, mode="r", shape=(n_test, 1))
    ref = []
    for s in range(0, n_test, T):
        e = min(n_test, s + T)
        one = {"neighbor_idxs": blk["ids"][s:e], "tgt_feats": blk["tgt_feats"][s:e], "targets": blk["targets"][s:e],
               "knn_sims": None, "knn_ids": None}
        ref.append(pipeline.eval_block(one, c["model"], 0.0, 1.0)["gcn_feat"].float().numpy())
    ref = np.concatenate(ref)
    assert np.array_equal(vals[:, 0], blk["targets"])
    assert np.abs(np.asarray(keys, np.float32) - ref).max() < (2e-2 if fp16 else 1e-4)       # fp16 rounding of O(1) features
    ds = DataStore.from_pretrained(str(sdir), use_memory=True)
    assert ds.dstore_size == n_test and ds.hidden_size == d and np.array_equal(np.asarray(ds.vals).reshape(-1), blk["targets"])
    with pytest.raises(ValueError):
        eval_lm.cli_main(c["base"] + ["--save-knnlm-dstore", "--dstore-mmap", str(out), "--knnlm", "--lmbda", "0.25",
                                      "--dstore-dir", str(c["data"] / "train_dstore")])


def test_eval_lm_train_split_invalid_neighbor_context(dev, tmp_path):
    """`--gen-subset train --save-knnlm-dstore --knn-keytype gcn_feat` (the datastore the kNN index of the GNN features is
    built over: gnnlm_scripts/wiki103/find_knn.sh:7): on the train split neighbours within --invalid-neighbor-context
    positions of their own token are dropped (token_block_dataset.py:360-362, language_modeling.py:299).  Keys == the
    oracle with the filter; the same run on the test split (filter off) and the oracle without it differ."""
    from gnnlm_amd import eval_lm, ops
    from oracle import pipeline
    c = make_data_dir(tmp_path)
    data, T, n_train, kg, ctx = c["data"], c["T"], c["n_train"], 6, 10
    rs = np.random.RandomState(11)
    pos = np.arange(n_train, dtype=np.int64)
    nb = rs.randint(0, n_train, size=(n_train, kg)).astype(np.int64)
    near = rs.random_sample(nb.shape) < 0.5                                   # half of the neighbours sit around the token itself
    nb = np.where(near, np.clip(pos[:, None] + rs.randint(-2 * ctx, 2 * ctx + 1, size=nb.shape), 0, n_train - 1), nb)
    nb[rs.random_sample(nb.shape) < 0.02] = -1
    nb[5] = pos[5] + np.arange(kg) - 2                                        # every neighbour inside the window: no star edges left
    nb.tofile(str(data / "train_dstore" / f"neighbors.mmap.{kg}"))
    train_vals = np.maximum(c["prob"]["vals"], 4).astype(np.int32)           # ids 0-3 are fairseq's specials (pad is stripped by the scorer)
    train_vals.astype(np.int16).tofile(str(data / "train_dstore" / "vals.npy"))
    # the HIP id rewrite == the rule, on the whole table
    got = ops.filter_neighbors(torch.from_numpy(nb).to(dev), torch.from_numpy(pos).to(dev), ctx).cpu().numpy()
    want = np.where((nb != -1) & (np.abs(pos[:, None] - nb) < ctx), -1, nb)
    assert np.array_equal(got, want) and (want != nb).mean() > 0.2
    same = ops.filter_neighbors(torch.from_numpy(nb).to(dev), torch.from_numpy(pos).to(dev), 0).cpu().numpy()
    assert np.array_equal(same, nb)
    base = list(c["base"])
    base[base.index("--gen-subset") + 1] = "train"
    out = tmp_path / "dstores"
    n_blocks = 4
    res = eval_lm.cli_main(base + ["--invalid-neighbor-context", str(ctx), "--first", str(n_blocks), "--save-knnlm-dstore",
                                   "--dstore-mmap", str(out)])
    n_tok = n_blocks * T
    assert res["count"] == n_tok
    d = c["prob"]["d"]
    keys = np.array(np.memmap(out / "train_dstore-gcn_feat" / "keys.npy", dtype=np.float32, mode="r", shape=(n_train, d))[:n_tok])
    vals = np.array(np.memmap(out / "train_dstore-gcn_feat" / "vals.npy", dtype=np.int32, mode="r", shape=(n_train, 1))[:n_tok, 0])
    train_keys = c["train_keys"]
    assert np.array_equal(vals, train_vals[:n_tok])
    ref, ref_nofilter = [], []
    for s in range(0, n_tok, T):
        one = {"neighbor_idxs": nb[s:s + T], "tgt_feats": train_keys[s:s + T], "targets": train_vals[s:s + T].astype(np.int64),
               "knn_sims": None, "knn_ids": None, "tgt_offsets": pos[s:s + T]}
        ref.append(pipeline.eval_block(one, dict(c["model"], invalid_neighbor_context=ctx), 0.0, 1.0)["gcn_feat"].float().numpy())
        ref_nofilter.append(pipeline.eval_block(one, c["model"], 0.0, 1.0)["gcn_feat"].float().numpy())
    ref, ref_nofilter = np.concatenate(ref), np.concatenate(ref_nofilter)
    assert np.abs(keys - ref).max() < 1e-4
    assert np.abs(ref - ref_nofilter).max() > 1e-2                            # the filter changes the features: the test bites


def test_eval_lm_word_outputs(dev, tmp_path, caplog):
    """--output-word-probs / --output-word-stats / --output-knn-recall / --remove-bpe (fairseq_cli/eval_lm.py:146-160,246-313,
    333-336): one line per hypothesis `<sample id> <word> [<score>] [<recall>]\\t...`, per-word statistics whose totals are the
    run's score sum and token count, BPE continuation tokens folded into the following token (count = tokens - skipped)."""
    import logging
    from gnnlm_amd import eval_lm
    c = make_data_dir(tmp_path)
    data, T, n_test, blk = c["data"], c["T"], c["n_test"], c["blk"]
    V = 600
    syms = [f"w{i}" + ("@@" if i % 5 == 0 else "") for i in range(4, V)]
    with open(data / "dict.txt", "w") as f:
        for s_ in syms:
            f.write(f"{s_} 1\n")
    knn = ["--knnlm", "--k", "8", "--lmbda", "0.25", "--dstore-dir", str(data / "train_dstore"), "--index-file",
           str(data / "train_dstore" / "faiss_store.cosine"), "--temperature", "1.0", "--knn-sim-func", "ip"]
    base1 = list(c["base"])
    base1[base1.index("--max-tokens") + 1] = str(T)
    plain = eval_lm.cli_main(base1 + knn)
    with caplog.at_level(logging.INFO, logger="gnnlm_amd.eval_lm"):
        res = eval_lm.cli_main(base1 + knn + ["--output-word-probs", "--output-word-stats", "--output-knn-recall"])
    assert res["count"] == n_test and abs(res["score_sum"] - plain["score_sum"]) < 1e-6 * n_test
    lines = [r.getMessage() for r in caplog.records if "\t" in r.getMessage() and "[" in r.getMessage()]
    n_blocks = -(-n_test // T)
    assert len(lines) == n_blocks
    first = lines[0].split(" ", 1)
    assert first[0] == "0"                                                    # sample id = block index in the split
    words = first[1].split("\t")
    assert len(words) == T
    tok0 = int(blk["targets"][0])
    assert words[0].startswith(("w%d" % tok0) + ("@@" if tok0 % 5 == 0 else "") + " [")
    assert words[0].count("[") == 2                                           # score and recall
    assert lines[-1].split(" ", 1)[0] == str(n_blocks - 1)
    ws = res["word_stats"]
    assert sum(w.count for w in ws.values()) == n_test
    assert abs(sum(w.log_prob for w in ws.values()) - res["score_sum"]) < 1e-3
    some = next(iter(ws.values()))
    assert str(some).count("\t") == 5
    # --remove-bpe: tokens whose symbol ends with the continuation marker are folded into the next one
    caplog.clear()
    with caplog.at_level(logging.INFO, logger="gnnlm_amd.eval_lm"):
        bpe = eval_lm.cli_main(base1 + ["--output-word-stats", "--remove-bpe"])
    tg = blk["targets"][:n_test]
    skipped = 0
    for s in range(0, n_test, T):
        seg = tg[s:s + T]
        skipped += int(((seg[:-1] % 5) == 0).sum())                           # (the last token of a hypothesis is never folded, :258)
    assert bpe["count"] == n_test - skipped
    assert not any(k_.endswith("@@") for k_ in bpe["word_stats"])            # continuation markers are stripped from the words
    # without dict.txt words are printed as token ids
    (data / "dict.txt").unlink()
    caplog.clear()
    with caplog.at_level(logging.INFO, logger="gnnlm_amd.eval_lm"):
        eval_lm.cli_main(base1 + ["--output-word-probs"])
    l0 = [r.getMessage() for r in caplog.records if "\t" in r.getMessage() and "[" in r.getMessage()][0]
    assert l0.split(" ", 1)[1].split("\t")[0].startswith(f"{tok0} [")
    with pytest.raises(ValueError):
        eval_lm.cli_main(base1 + ["--remove-bpe"])
    with pytest.raises(ValueError):
        eval_lm.cli_main(base1 + ["--output-knn-recall", "--output-word-probs"])


def test_quantize_features_producer(dev, tmp_path):
    """`python -m gnnlm_amd.quantize_features` (knn/quantize_features.py:29-157): strided sample -> OPQ + PQ trained on the GPU ->
    `quantizer` in faiss's serialisation + `quantized-keys.npy` as a real .npy; the codes are the oracle's encode under the trained
    codebook, the logged error is the reference's definition, OPQ training beats the untrained rotation, `eval_quantizer` and the
    eval driver read the outputs."""
    from gnnlm_amd import eval_quantizer, quantize_features
    from gnnlm_amd.faiss_io import read_pq_quantizer
    from gnnlm_amd.pq_wrapper import TorchPQCodec
    from oracle import pq as opq_
    rs = np.random.RandomState(11)
    N, d, M = 41_000, 64, 16
    # anisotropic, correlated keys: a rotation that balances the sub-spaces matters
    z = rs.randn(N, d).astype(np.float32) * (np.linspace(3.0, 0.05, d) ** 1)[None, :]
    mix = np.linalg.qr(rs.randn(d, d))[0].astype(np.float32)
    keys = (z @ mix + 0.3 * rs.randn(1, d)).astype(np.float16)
    data = tmp_path / "data-bin"
    write_dstore(str(data / "train_dstore"), keys, rs.randint(4, 500, N).astype(np.int16), 500)
    argv = ["--data-dir", str(data), "--subset", "train", "--index", f"OPQ{M}_{d},,PQ{M}", "--code-size", str(M), "--chunk-size", "20000",
            "--compute-error", "--pq-iters", "12", "--opq-iters", "8", "--encode-rows", "16384"]
    out = quantize_features.main(quantize_features.get_parser().parse_args(argv))
    codes = np.load(str(data / "train_dstore" / "quantized-keys.npy"))                    # language_modeling.py:276
    assert codes.shape == (N, M) and codes.dtype == np.uint8 and out["quantized_keys"].endswith("quantized-keys.npy")
    q = read_pq_quantizer(str(data / "quantizer"))
    cen, A, b = q["centroids"], q["A"], q["b"]
    assert cen.shape == (M, 256, d // M) and A.shape == (d, d) and b.size == 0
    np.testing.assert_allclose(A @ A.T, np.eye(d), atol=1e-4)                            # an orthonormal rotation
    x = keys.astype(np.float32)
    ref = opq_.pq_encode(x, cen, A, b)
    assert (codes != ref).mean() < 1e-3                                                   # near-ties aside, the oracle's codes
    dec = opq_.pq_decode(codes, cen, A, b)
    per_batch = [(((x[s:s + 8192] - dec[s:s + 8192]) ** 2).sum() / (x[s:s + 8192] ** 2).sum()) * len(x[s:s + 8192]) for s in range(0, N, 8192)]
    assert abs(out["avg_reconstruction_error"] - sum(per_batch) / N) < 1e-4 * out["avg_reconstruction_error"] + 1e-7   # :139-148
    err = eval_quantizer.main(eval_quantizer.get_parser().parse_args(["--data-dir", str(data)]))
    per_row = (((x - dec) ** 2).sum(1) / (x ** 2).sum(1)).mean()
    assert abs(err - per_row) < 1e-4 * per_row + 1e-7                                     # eval_quantizer.py:62-66
    # the untrained (random) rotation is worse; --pretrained_quantizer re-encodes with the stored quantizer and reproduces the codes
    out0 = quantize_features.main(quantize_features.get_parser().parse_args([a if a != "8" else "0" for a in argv]))   # --opq-iters 0
    assert out0["avg_reconstruction_error"] > 1.02 * out["avg_reconstruction_error"]
    write_q = quantize_features.write_pq_quantizer
    write_q(str(data / "quantizer"), cen, A, b, metric="l2")
    quantize_features.main(quantize_features.get_parser().parse_args(argv + ["--pretrained_quantizer"]))
    assert np.array_equal(np.load(str(data / "train_dstore" / "quantized-keys.npy")), codes)
    # the codec the hot path builds from the file decodes them like the oracle
    c = TorchPQCodec.from_file(str(data / "quantizer")).to(dev)
    got = c.decode(torch.from_numpy(codes[:3000]).to(dev)).cpu().numpy()
    np.testing.assert_allclose(got, dec[:3000], atol=2e-5, rtol=1e-5)
    # a reducing block (one_billion/find_knn.sh:24 `OPQ128_512`): A [d_out, d_in] with orthonormal rows
    argv_r = [a for a in argv]
    argv_r[argv_r.index(f"OPQ{M}_{d},,PQ{M}")] = f"OPQ{M // 2}_{d // 2},,PQ{M // 2}"
    argv_r[argv_r.index("--code-size") + 1] = str(M // 2)
    out_r = quantize_features.main(quantize_features.get_parser().parse_args(argv_r))
    q_r = read_pq_quantizer(str(data / "quantizer"))
    assert q_r["A"].shape == (d // 2, d) and q_r["centroids"].shape == (M // 2, 256, d // M)
    assert np.load(str(data / "train_dstore" / "quantized-keys.npy")).shape == (N, M // 2)
    np.testing.assert_allclose(q_r["A"] @ q_r["A"].T, np.eye(d // 2), atol=1e-4)
    assert out_r["avg_reconstruction_error"] < 0.4                              # (8-byte codes of 64-d keys, half the dimensions dropped)


def test_eval_lm_reads_the_producers_outputs(dev, tmp_path):
    """quantize_features -> eval_lm: the driver scores a split over a store whose codes and quantizer file came from the producer
    (checkpoint without tgt_quantizer buffers, `quantizer_path` override as in the recipe, hgt_lm_wiki103_reproduce.sh:86), and the
    ppl equals the oracle's over the same codes."""
    from gnnlm_amd import eval_lm, quantize_features
    from gnnlm_amd.faiss_io import read_pq_quantizer
    from oracle import pipeline
    c = make_data_dir(tmp_path)
    data, base, model, T, n_test, blk = c["data"], c["base"], c["model"], c["T"], c["n_test"], c["blk"]
    quantize_features.main(quantize_features.get_parser().parse_args(
        ["--data-dir", str(data), "--subset", "train", "--index", "OPQ16_64,,PQ16", "--code-size", "16", "--chunk-size", "2000",
         "--pq-iters", "5", "--opq-iters", "2"]))
    q = read_pq_quantizer(str(data / "quantizer"))
    ck = torch.load(str(tmp_path / "ckpt.pt"), weights_only=False)
    for k_ in [k_ for k_ in ck["model"] if "tgt_quantizer" in k_]:
        del ck["model"][k_]
    torch.save(ck, str(tmp_path / "ckpt.pt"))
    base = list(base)
    base[base.index("--model-overrides") + 1] = "{'orig_prob_ratio': 0.0, 'quantizer_path': '%s'}" % str(data / "quantizer")
    res = eval_lm.cli_main(base)
    model = dict(model, centroids=q["centroids"], A=q["A"], b=q["b"], codes=np.load(str(data / "train_dstore" / "quantized-keys.npy")))
    total = 0.0
    for s in range(0, n_test, T):
        e = min(n_test, s + T)
        one = {"neighbor_idxs": blk["ids"][s:e], "tgt_feats": blk["tgt_feats"][s:e], "targets": blk["targets"][s:e], "knn_sims": None, "knn_ids": None}
        total += pipeline.eval_block(one, model, 0.0, 1.0)["lm_logp"].double().sum().item()
    assert res["count"] == n_test and abs(res["score_sum"] - total) < 1e-4 * n_test


@pytest.mark.parametrize("metric_type", ["ip", "l2"])
def test_knn_model_recompute_from_host_keys(dev, golden, tmp_path, metric_type):
    """`--knn-sim-func ip | l2` with a key table that does not go to HBM (knn_model.py:163,170 gather `self.keys[knns]` from the
    np.memmap): rows gathered from the host per query block, arithmetic on the device -- the numbers of the in-HBM path and the
    reference's own (golden `knn.npz`)."""
    from gnnlm_amd.knn_model import KNNModel
    g = golden("knn")
    d = str(tmp_path / "train_dstore")
    write_dstore(d, g["keys"], g["vals"].astype(np.int16), 50)
    q = torch.from_numpy(g["queries"]).to(dev)
    tg = torch.from_numpy(g["targets"]).to(dev)
    for cosine in (False, True):
        tag = f"{metric_type}.{'cos' if cosine else 'raw'}.t1.0"
        idx_file = "faiss_store.cosine" if cosine else "faiss_store.ip"
        outs = []
        for bound, blk in ((float("inf"), 1 << 30), (0, 1 << 30), (0, 3 * 8 * g["keys"].shape[1] * 4)):     # in HBM / host gather / tiny blocks
            m = KNNModel(idx_file, d, metric_type=metric_type, k=8, index=FixedIndex(g[tag + ".dists"], g[tag + ".ids"]), device=dev)
            m.max_hbm_key_bytes, m.host_gather_bytes = bound, blk
            p, rec = m.get_knn_prob(q, t=1.0, targets=tg, return_recall=True)
            assert (m._keys_device() is None) == (bound == 0)
            outs.append((p.cpu().numpy(), rec.cpu().numpy()))
        for p_, r_ in outs[1:]:
            np.testing.assert_allclose(p_, outs[0][0], rtol=1e-6, atol=1e-9)
            assert np.array_equal(r_, outs[0][1])
        assert np.array_equal(outs[0][1], g[tag + ".recall"])
        np.testing.assert_allclose(outs[1][0], g[tag + ".p"], rtol=3e-5, atol=1e-7)


def test_pipeline_from_raw_keys(dev, tmp_path):
    """The whole pipeline of the recipes from nothing but raw key tables, every producer this repo's own (no faiss):
    quantize_features (find_knn.sh:32-38) -> run_index_build (find_knn.sh:8-13) -> find_knn (:17-22) -> eval_lm --graph --knnlm
    (hgt_lm_wiki103_reproduce.sh:140-150).  The driver's score equals the oracle's over the SAME produced files (codes, quantizer,
    neighbour ids, index arrays)."""
    from gnnlm_amd import eval_lm, find_knn, quantize_features, run_index_build
    from gnnlm_amd.faiss_io import read_pq_quantizer
    from gnnlm_amd.synthetic import make_problem
    from oracle import ivfpq as oivf, knn as oknn_, pipeline
    d, H, M, dsub, V, kg, T, L, k = 64, 4, 16, 4, 600, 6, 16, 2, 32
    n_train, n_test = 6000, 41
    prob = make_problem(n_store=n_train, d=d, n_heads=H, M=M, dsub=dsub, vocab=V, cutoff=[100, 300], T=n_test, kg=kg,
                        left=2, right=2, n_layers=L, k=8, seed=5)
    rs = np.random.RandomState(9)
    centres = rs.randn(30, d).astype(np.float32)
    train_keys = (centres[rs.randint(0, 30, n_train)] + 0.5 * rs.randn(n_train, d)).astype(np.float16)
    test_keys = (centres[rs.randint(0, 30, n_test)] + 0.5 * rs.randn(n_test, d)).astype(np.float16)
    targets = np.maximum(prob["block"]["targets"], 4)
    data = tmp_path / "data-bin"
    write_dstore(str(data / "train_dstore"), train_keys, prob["vals"].astype(np.int16), V)
    write_dstore(str(data / "test_dstore"), test_keys, targets.astype(np.int16), V)
    # 1. the quantizer and the code table
    quantize_features.main(quantize_features.get_parser().parse_args(
        ["--data-dir", str(data), "--subset", "train", "--index", f"OPQ{M}_{d},,PQ{M}", "--code-size", str(M), "--chunk-size", "6000",
         "--pq-iters", "6", "--opq-iters", "3"]))
    # 2. the kNN index over the train keys, 3. the graph neighbours of the test split (searched over that index)
    run_index_build.main(run_index_build.get_parser().parse_args(
        ["--dstore-dir", str(data / "train_dstore"), "--index-type", f"OPQ{M}_{d},IVF32,PQ{M}", "--metric", "cosine", "--nprobe", "8"]))
    find_knn.main(find_knn.get_parser().parse_args(["--data-dir", str(data), "--subset", "test", "--k", str(kg), "--nprobe", "8"]))
    nbrs = np.array(np.memmap(str(data / "test_dstore" / f"neighbors.mmap.{kg}"), dtype=np.int64, mode="r", shape=(n_test, kg)))
    assert nbrs.min() >= 0 and nbrs.max() < n_train
    # 4. the driver, with a checkpoint that carries no quantizer (the recipe's `quantizer_path` override)
    sd = {"decoder.hgt_decoder." + k_: v for k_, v in prob["sd"].items()}
    w = prob["asm"]
    for i, e in enumerate(w["emb"]):
        sd[f"decoder.embed_tokens.embeddings.{i}.0.weight"] = e
        if i:
            sd[f"decoder.embed_tokens.embeddings.{i}.1.weight"] = w["proj"][i]
    sd["decoder.adaptive_softmax.head.class_proj.weight"] = w["class_proj"]
    margs = Namespace(decoder_embed_dim=d, decoder_attention_heads=H, graph_layer=L, decoder_gcn_dim=d,
                      adaptive_softmax_cutoff="100,300", orig_prob_ratio=0.0, short_cut=False, quantizer_path="")
    torch.save({"args": margs, "model": sd}, str(tmp_path / "ckpt.pt"))
    lam, temp = 0.25, 1.0
    cmd = [str(data), "--path", str(tmp_path / "ckpt.pt"), "--gen-subset", "test", "--graph", "--neighbor-context", "2", "--gcn-k", str(kg),
           "--use-precompute-feat", "--sample-break-mode", "none", "--max-tokens", str(T), "--tokens-per-sample", str(T),
           "--gcn-context-window", "0", "--knn-keytype", "gcn_feat", "--model-overrides",
           "{'orig_prob_ratio': 0.0, 'quantizer_path': '%s'}" % str(data / "quantizer"),
           "--knnlm", "--k", str(k), "--lmbda", str(lam), "--dstore-dir", str(data / "train_dstore"),
           "--index-file", str(data / "train_dstore" / "faiss_store.cosine"), "--temperature", str(temp), "--knn-sim-func", "do_not_recomp_ip", "--probe", "8"]
    res = eval_lm.cli_main(cmd)
    # the recipe's literal one-block batches, in turn on six streams (the driver's default for them), each with its own IVF-PQ search
    # in flight (two-phase search: pinned landing slots, events): the same scores
    lit = eval_lm.cli_main(cmd + ["--batch-blocks", "0"])
    assert lit["count"] == res["count"] and abs(lit["score_sum"] - res["score_sum"]) <= 1e-9 * abs(res["score_sum"])
    # the oracle over the produced files
    q = read_pq_quantizer(str(data / "quantizer"))
    z = np.load(str(data / "train_dstore" / "faiss_store.cosine.gnnlm.npz"))
    model = {"sd": prob["sd"], "n_layers": L, "n_heads": H, "centroids": q["centroids"], "A": q["A"], "b": q["b"],
             "codes": np.load(str(data / "train_dstore" / "quantized-keys.npy")), "vals": prob["vals"], "n_store": n_train,
             "left": 2, "right": 2, "asm": w}
    total = 0.0
    for s in range(0, n_test, T):
        e = min(n_test, s + T)
        one = {"neighbor_idxs": nbrs[s:e], "tgt_feats": test_keys[s:e], "targets": targets[s:e], "knn_sims": None, "knn_ids": None}
        o = pipeline.eval_block(one, model, 0.0, 1.0)
        qn = oknn_.normalize_queries(o["gcn_feat"].float(), True).numpy()
        dd, ii = oivf.search(qn, z["R"], z["coarse"], z["pq"], z["list_off"], z["list_ids"], z["list_codes"], k=k, nprobe=8)
        p, _ = oknn_.knn_target_prob(dd.astype(np.float32), ii, prob["vals"], targets[s:e], temp)
        total += oknn_.combine_knn_and_vocab_probs(p, o["lm_logp"], lam).double().sum().item()
    assert res["count"] == n_test and abs(res["score_sum"] - total) < 5e-4 * n_test


@pytest.mark.parametrize("L", [1, 3])
def test_eval_lm_multi_process_sharded_store(dev, tmp_path, L):
    """`python -m torch.distributed.run --nproc-per-node 2 -m gnnlm_amd.eval_lm ...` (both ranks on device 0, collectives staged
    through the host -- RCCL refuses two ranks per device): the driver initialises the process group itself, every rank scores
    its share of the blocks against a range-sharded (exchange, exact and fixed-capacity), peer-mapped or replicated code table,
    one all-reduce at the end, rank 0 prints the reference's two lines -- the same lines, and the same score sum to the last
    bits float64 addition order allows, as the single-process run.  L = 3 merges equal context groups BEFORE the exchange
    (fewer bytes on the links than groups x rows); one rank has a batch less than the other and serves its peer meanwhile."""
    import json
    import subprocess
    import sys
    c = make_data_dir(tmp_path, n_test=100, L=L)                            # 6 full blocks + a 4-token one: 4 blocks / 3 blocks
    base = list(c["base"])
    base[base.index("--max-tokens") + 1] = str(c["T"])                     # the recipe's one-block batches
    knn = ["--knnlm", "--k", "8", "--lmbda", "0.25", "--dstore-dir", str(c["data"] / "train_dstore"),
           "--index-file", str(c["data"] / "train_dstore" / "faiss_store.cosine"), "--temperature", "1.0", "--knn-sim-func", "ip"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")

    def run(extra, ranks, port):
        out = str(tmp_path / f"res_{port}.json")
        cmd = [sys.executable] + (["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
                                   "--master-port", str(port)] if ranks > 1 else []) + ["-m", "gnnlm_amd.eval_lm"] + base + knn + extra + ["--result-json", out]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root,
                           env=dict(env, GNNLM_EVAL_BACKEND="gloo", GNNLM_EVAL_DEVICE="0"))
        assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
        lines = [l for l in p.stdout.splitlines() if l.startswith(("Evaluated", "Loss"))]
        assert len(lines) == 2, p.stdout                                    # rank 0 prints, once
        return lines, json.load(open(out)), [json.load(open(out + f".rank{r}")) for r in range(ranks)] if ranks > 1 else None
    one_lines, one, _ = run([], 1, 0)
    assert one["count"] == 100 and one["world"] == 1
    bytes_seen = {}
    for i, extra in enumerate([["--store", "sharded"], ["--store", "sharded", "--exchange", "padded"], ["--store", "peer"], ["--store", "replicated"]]):
        lines, res, ranks = run(extra, 2, 29650 + 4 * L + i)
        assert lines[1] == one_lines[1], (extra, lines, one_lines)          # "Loss (base 2): ..., Perplexity: ..." byte for byte
        assert res["count"] == 100 and res["tokens"] == 100 and res["world"] == 2 and res["store"] == extra[1]
        assert abs(res["score_sum"] - one["score_sum"]) <= 1e-12 * abs(one["score_sum"])
        assert [r["rank_tokens"] for r in ranks] == [64, 36]
        assert abs(sum(r["rank_score_sum"] for r in ranks) - one["score_sum"]) <= 1e-12 * abs(one["score_sum"])
        if extra[1] == "sharded":
            assert all(r["xgmi_bytes"] > 0 for r in ranks)
            bytes_seen[tuple(extra)] = ranks[0]["xgmi_bytes"]
    if L > 1:
        # requests were merged before the exchange: rank 0 (64 tokens x 6 neighbours) asked for fewer groups than it has neighbours
        M, n_g = 16, 5
        assert bytes_seen[("--store", "sharded")] < 64 * 6 * (8 + n_g * M)
