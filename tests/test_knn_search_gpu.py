"""On-device kNN search: the running top-k merge kernel and the chunked exact index against a
stable argsort / the oracle's brute-force search (knn/knn_model.py:87-101 contract: best first, -1 padding)."""
import struct

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import knn as oknn


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def ref_topk(vals, ids, k, largest):
    """Stable selection: better value first, ties by ascending id."""
    key = -vals if largest else vals
    order = np.lexsort((ids, key), axis=-1) if vals.ndim == 1 else np.stack([np.lexsort((ids[r], key[r])) for r in range(len(vals))])
    top = order[..., :k]
    return np.take_along_axis(vals, top, -1), np.take_along_axis(ids, top, -1)


@pytest.mark.parametrize("k", [8, 70, 256, 1024, 1500])
@pytest.mark.parametrize("largest", [True, False])
def test_topk_merge_chunked(dev, k, largest):
    from gnnlm_amd import ops
    rs = np.random.RandomState(k + largest)
    n, N = 37, 9000
    scores = rs.randn(n, N).astype(np.float32)
    scores[:, ::50] = scores[:, 1::50]                                   # exact ties across columns
    scale = rs.uniform(0.5, 2.0, N).astype(np.float32)
    bias = rs.randn(N).astype(np.float32)
    alpha = -2.0 if not largest else 0.7
    val = np.float32(alpha) * scores * scale[None] + bias[None]
    ids = np.tile(np.arange(N, dtype=np.int64) + 1000, (n, 1))
    v_ref, i_ref = ref_topk(val, ids, k, largest)
    bv = torch.empty(n, k, device=dev)
    bi = torch.empty(n, k, device=dev, dtype=torch.int64)
    s_dev = torch.from_numpy(scores).to(dev)
    edges = [0, 1, 700, 701, 4096, 8999, N]                               # ragged chunks, incl. 1-column ones
    for a, b in zip(edges[:-1], edges[1:]):
        ops.topk_merge(s_dev[:, a:b], bv, bi, col0=1000 + a, col_scale=torch.from_numpy(scale[a:b]).to(dev),
                       col_bias=torch.from_numpy(bias[a:b]).to(dev), alpha=alpha, largest=largest, init=(a == 0))
    assert np.array_equal(bi.cpu().numpy(), i_ref)                        # ids: exact, ties by ascending id
    np.testing.assert_allclose(bv.cpu().numpy(), v_ref, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("k", [70, 1024])
@pytest.mark.parametrize("largest", [True, False])
def test_topk_merge_wide_first_chunk(dev, k, largest):
    """One wide first chunk (the dense round of an IVF search): the counting pre-pass (value histogram -> admitted bins)
    may not change the selection -- exact ids with ties by ascending id, heavy ties at the k-th place, a constant row, a
    row with fewer valid columns than k, values far outside the sampled range."""
    from gnnlm_amd import ops
    rs = np.random.RandomState(7 * k + largest)
    n, N = 9, 20000
    scores = rs.randn(n, N).astype(np.float32)
    scores[:, ::40] = scores[:, 1::40]                                   # exact ties across columns
    scores[1] = np.round(scores[1] * 2) / 2                               # a few distinct values: thousands of ties at the k-th place
    scores[2] = 0.25                                                      # constant row
    scores[3, 5000:] *= 1000.0                                            # the first 4096 columns (the sampled range) miss the tail
    col_ids = np.arange(N, dtype=np.int64) + 7
    ncols = np.full(n, N, dtype=np.int32)
    ncols[4] = k // 2                                                     # fewer columns than k
    bv = torch.empty(n, k, device=dev)
    bi = torch.empty(n, k, device=dev, dtype=torch.int64)
    ops.topk_merge(torch.from_numpy(scores).to(dev), bv, bi, col_ids=torch.from_numpy(col_ids).to(dev), largest=largest, init=True,
                   row_ncols=torch.from_numpy(ncols).to(dev))
    bv, bi = bv.cpu().numpy(), bi.cpu().numpy()
    for r in range(n):
        m = min(k, int(ncols[r]))
        v, i = ref_topk(scores[r, :ncols[r]], col_ids[:ncols[r]], k, largest)
        assert np.array_equal(bi[r, :m], i[:m]), r
        np.testing.assert_array_equal(bv[r, :m], v[:m])
        assert (bi[r, m:] == -1).all()


def test_topk_merge_fewer_than_k_and_explicit_ids(dev):
    from gnnlm_amd import ops
    rs = np.random.RandomState(3)
    n, N, k = 5, 40, 64
    scores = rs.randn(n, N).astype(np.float32)
    col_ids = rs.permutation(10_000_000_000 + np.arange(N)).astype(np.int64)     # ids beyond int32
    col_ids[7] = -1                                                       # skipped column
    ncols = np.array([40, 0, 13, 40, 1], dtype=np.int32)                  # ragged rows
    bv = torch.full((n, k), 123.0, device=dev)
    bi = torch.full((n, k), 77, device=dev, dtype=torch.int64)
    ops.topk_merge(torch.from_numpy(scores).to(dev), bv, bi, col_ids=torch.from_numpy(col_ids).to(dev), largest=True, init=True,
                   row_ncols=torch.from_numpy(ncols).to(dev))
    bv, bi = bv.cpu().numpy(), bi.cpu().numpy()
    for r in range(n):
        ok = (np.arange(N) < ncols[r]) & (col_ids >= 0)
        v, i = ref_topk(scores[r][ok], col_ids[ok], k, True)
        m = ok.sum()
        assert np.array_equal(bi[r, :m], i) and (bi[r, m:] == -1).all()   # faiss pads with -1
        np.testing.assert_allclose(bv[r, :m], v)
        assert np.isneginf(bv[r, m:]).all()


@pytest.mark.parametrize("metric,cosine", [("ip", False), ("ip", True), ("l2", False)])
def test_exact_index_chunked_fp16_store(dev, metric, cosine):
    """200k fp16 keys searched in 7 chunks x 2 query blocks == the oracle's brute force over the whole table."""
    from gnnlm_amd.knn_model import ExactIndex
    rs = np.random.RandomState(5)
    N, d, n, k = 200_000, 64, 300, 100
    keys = rs.randn(N, d).astype(np.float16)
    q = rs.randn(n, d).astype(np.float32)
    idx = ExactIndex(keys, metric, cosine, dev, chunk_rows=30_000, score_bytes=4 * 30_000 * 160)
    assert idx.keys.dtype == torch.float16                                # read in place, no f32 copy of the table
    d_got, i_got = idx.search(q, k)
    d_ref, i_ref = oknn.brute_force_search(q, keys, k, metric, cosine)
    # float32 scores of near-equal candidates may swap ranks: compare as sets with a small allowance, values tightly
    same = np.mean([len(set(a) & set(b)) / k for a, b in zip(i_got, i_ref)])
    assert same > 0.999, same
    np.testing.assert_allclose(d_got, d_ref, rtol=2e-4, atol=2e-4)
    assert (np.diff(d_got, axis=1) <= 1e-6).all() if metric == "ip" else (np.diff(d_got, axis=1) >= -1e-6).all()


def test_exact_index_small_store_pads(dev):
    from gnnlm_amd.knn_model import ExactIndex
    rs = np.random.RandomState(1)
    keys = rs.randn(5, 16).astype(np.float32)
    d, i = ExactIndex(keys, "ip", False, dev).search(rs.randn(3, 16).astype(np.float32), 8)
    assert (i[:, 5:] == -1).all() and (np.sort(i[:, :5], 1) == np.arange(5)).all()


def test_ivfpq_search_matches_oracle(dev):
    """IVF-PQ ADC search on the device (coarse GEMM -> probes -> LUT GEMM -> list scan in two rounds -> top-k) against
    the numpy restatement of IVFADC over the same index arrays: 1 M synthetic keys, d = 256, OPQ + IVF256 + PQ16,
    nprobe 12, k = 1024 and 100 (the reference's k=1024 search, knn_model.py:100)."""
    from gnnlm_amd.ivfpq import IVFPQIndex
    from oracle import ivfpq as oivf
    rs = np.random.RandomState(9)
    N, d, nlist, M = 1_000_000, 256, 256, 16
    # clustered keys (an index over pure noise has nothing to find): 500 centres + noise, fp16 like keys.npy
    centres = rs.randn(500, d).astype(np.float32)
    keys = (centres[rs.randint(0, 500, N)] + 0.7 * rs.randn(N, d).astype(np.float32)).astype(np.float16)
    index = IVFPQIndex.build(keys, nlist, M, device=dev, cosine=True, nprobe=12, iters=6, seed=1)
    assert index.list_codes.shape == (N, M) and int(index.list_off[-1]) == N
    q = (centres[rs.randint(0, 500, 40)] + 0.7 * rs.randn(40, d)).astype(np.float32)
    qn = q / np.sqrt((q ** 2).sum(1, keepdims=True))                      # KNNModel normalises the queries (knn_model.py:181-184)
    arrs = [getattr(index, a).cpu().numpy() for a in ("R", "coarse", "pq", "list_off", "list_ids", "list_codes")]
    for k in (1024, 100):
        v, i = index.search(qn, k)
        v_ref, i_ref = oivf.search(qn, *arrs, k=k, nprobe=12)
        same = np.mean([len(set(a) & set(b)) / k for a, b in zip(i, i_ref)])
        assert same > 0.998, same                                         # float32 vs float64 near-ties at the k-th place
        np.testing.assert_allclose(v, v_ref, rtol=2e-5, atol=2e-5)
        assert (np.diff(v, axis=1) <= 0).all() and (i >= 0).all()
    # the index finds real neighbours: what it returns lies among the exact nearest ~0.2 % of the store (PQ16 cannot
    # rank inside a cluster of equidistant points, so this is a membership check, not a rank check)
    from gnnlm_amd.knn_model import ExactIndex
    _, ie = ExactIndex(keys, "ip", True, dev).search(qn, 2048)
    _, ia = index.search(qn, 100)
    assert np.mean([len(set(a) & set(b)) / 100 for a, b in zip(ia, ie)]) > 0.8
    # save / load round trip
    import os, tempfile
    with tempfile.TemporaryDirectory() as td:
        index.save(os.path.join(td, "ivfpq.npz"))
        again = IVFPQIndex.load(os.path.join(td, "ivfpq.npz"), device=dev)
        v2, i2 = again.search(qn, 100)
        assert np.array_equal(i2, ia)


@pytest.mark.parametrize("M", [64, 32])
def test_ivfpq_packed_scan_matches_oracle(dev, M):
    """M = 32 / 64 (the reference's PQ64): the scan over the packed image (rotated code bytes, [half][code][sub-quantizer]
    tables) against the numpy IVFADC over the plain index arrays; uneven lists, a list shorter than a 64-row block."""
    from gnnlm_amd import _lib
    from gnnlm_amd.ivfpq import IVFPQIndex
    from oracle import ivfpq as oivf
    rs = np.random.RandomState(21 + M)
    N, d, nlist = 150_001, 256, 24
    centres = rs.randn(60, d).astype(np.float32)
    keys = (centres[rs.randint(0, 60, N)] + 0.6 * rs.randn(N, d).astype(np.float32)).astype(np.float16)
    index = IVFPQIndex.build(keys, nlist, M, device=dev, cosine=True, nprobe=9, iters=5, seed=3, scan="f32")     # (M = 64 defaults to the int8-MFMA search: tests/test_ivfpq_mfma_gpu.py)
    assert index.packed_codes is not None and index.packed_codes.numel() == -(-N // 64) * 64 * M
    # the packed image, restated: block of 64 rows, piece-major, byte s of half h = sub-quantizer 32 h + (row + s) mod 32
    codes = index.list_codes.cpu().numpy()
    pk = index.packed_codes.cpu().numpy().reshape(-1, M // 16, 64, 16)
    for r in (0, 1, 31, 63, 64, 12345, N - 1):
        row = pk[r // 64, :, r % 64, :].reshape(M)
        want = np.array([codes[r, 32 * (i // 32) + (r + i % 32) % 32] for i in range(M)], dtype=np.uint8)
        assert np.array_equal(row, want), r
    assert not pk[-1, :, N % 64:, :].any()                                    # rows beyond N are zero
    q = (centres[rs.randint(0, 60, 33)] + 0.6 * rs.randn(33, d)).astype(np.float32)
    qn = q / np.sqrt((q ** 2).sum(1, keepdims=True))
    arrs = [getattr(index, a).cpu().numpy() for a in ("R", "coarse", "pq", "list_off", "list_ids", "list_codes")]
    for k in (1024, 64):
        v, i = index.search(qn, k)
        v_ref, i_ref = oivf.search(qn, *arrs, k=k, nprobe=9)
        same = np.mean([len(set(a) & set(b)) / k for a, b in zip(i, i_ref)])
        assert same > 0.998, same
        np.testing.assert_allclose(v, v_ref, rtol=2e-5, atol=2e-5)
        assert (np.diff(v, axis=1) <= 0).all() and (i >= 0).all()
    # too few candidate slots for the thresholded round: the search notices (survivor counts) and repeats itself with room
    tight = IVFPQIndex(index.R, index.coarse, index.pq, index.list_off, index.list_ids, index.list_codes, nprobe=9,
                       dense_probes=1, cand_cap=2, scan="f32")
    qr = rs.randn(9, d).astype(np.float32)                                   # unclustered queries: neighbours in every probed list
    qr /= np.sqrt((qr ** 2).sum(1, keepdims=True))
    vf, jf = index.search(qr, 1024)
    v3, i3 = tight.search(qr, 1024)
    assert tight.cand_cap > 2
    np.testing.assert_allclose(v3, vf, rtol=1e-5, atol=1e-5)
    assert np.mean([len(set(a) & set(b)) / 1024 for a, b in zip(jf, i3)]) > 0.998
    # the same index searched by the row-major kernels gives the same neighbours
    plain = IVFPQIndex(index.R, index.coarse, index.pq, index.list_off, index.list_ids, index.list_codes, nprobe=9, scan="f32")
    plain.packed_codes = None
    v2, i2 = plain.search(qn, 64)
    np.testing.assert_allclose(v2, v, rtol=1e-5, atol=1e-5)
    assert np.mean([len(set(a) & set(b)) / 64 for a, b in zip(i, i2)]) > 0.998


@pytest.mark.parametrize("d,M", [(64, 16), (128, 32), (256, 64)])
def test_ivfpq_small_lists_and_padding(dev, d, M):
    """Lists shorter than a 64-row block, fewer stored keys than k (-1 padding), an odd query count (a workgroup with one
    task); M = 32 / 64 take the packed-image kernels."""
    from gnnlm_amd.ivfpq import IVFPQIndex
    from oracle import ivfpq as oivf
    rs = np.random.RandomState(2)
    keys = rs.randn(300, d).astype(np.float32)
    index = IVFPQIndex.build(keys, 8, M, device=dev, cosine=False, nprobe=3, iters=4, seed=0)
    assert (index.packed_codes is not None) == (M == 32) and (index.tiles is not None) == (M == 64)
    q = rs.randn(5, d).astype(np.float32)
    v, i = index.search(q, 200)                                           # more than 3 lists hold: -1 padding
    arrs = [getattr(index, a).cpu().numpy() for a in ("R", "coarse", "pq", "list_off", "list_ids", "list_codes")]
    v_ref, i_ref = oivf.search(q, *arrs, k=200, nprobe=3)
    assert np.array_equal(i >= 0, i_ref >= 0)
    assert np.array_equal(np.sort(np.where(i >= 0, i, 10 ** 9), 1), np.sort(np.where(i_ref >= 0, i_ref, 10 ** 9), 1))
    ok = i >= 0
    np.testing.assert_allclose(v[ok], v_ref[ok], rtol=2e-5, atol=2e-5)


def test_knn_model_over_built_index(dev, tmp_path):
    """run_index_build -> KNNModel(index_file=<the reference's path>) -> get_knn_prob: the whole kNN half of the eval path
    with the search on the device (knn_model.py:87-101,179-217), against oracle search + oracle kNN prob on the same index."""
    import json, os
    from gnnlm_amd import run_index_build
    from gnnlm_amd.ivfpq import IVFPQIndex
    from gnnlm_amd.knn_model import KNNModel
    from oracle import ivfpq as oivf
    rs = np.random.RandomState(4)
    N, d, V, n, k = 20000, 64, 50, 33, 64
    centres = rs.randn(40, d).astype(np.float32)
    keys = (centres[rs.randint(0, 40, N)] + 0.5 * rs.randn(N, d)).astype(np.float16)
    vals = rs.randint(0, V, N).astype(np.int16)
    dd = tmp_path / "train_dstore"
    os.makedirs(dd)
    keys.tofile(dd / "keys.npy"); vals.tofile(dd / "vals.npy")
    json.dump({"dstore_size": N, "hidden_size": d, "vocab_size": V, "dstore_fp16": True, "val_size": 1}, open(dd / "info.json", "w"))
    out = run_index_build.main(run_index_build.get_parser().parse_args(
        ["--dstore-dir", str(dd), "--index-type", "OPQ16_64,IVF64,PQ16", "--metric", "cosine", "--nprobe", "8"]))
    assert out.endswith("faiss_store.cosine.gnnlm.npz")
    m = KNNModel(str(dd / "faiss_store.cosine"), str(dd), k=k, probe=8, no_load_keys=True, metric_type="do_not_recomp_ip", device=dev)
    assert isinstance(m.index, IVFPQIndex) and m.cosine
    q = (centres[rs.randint(0, 40, n)] + 0.5 * rs.randn(n, d)).astype(np.float32)
    targets = rs.randint(0, V, n).astype(np.int64)
    p, rec = m.get_knn_prob(torch.from_numpy(q).to(dev), targets=torch.from_numpy(targets).to(dev), t=1.0, return_recall=True)
    arrs = [getattr(m.index, a).cpu().numpy() for a in ("R", "coarse", "pq", "list_off", "list_ids", "list_codes")]
    v_ref, i_ref = oivf.search(q, *arrs, k=k, nprobe=8, cosine_queries=True)
    p_ref, rec_ref = oknn.knn_target_prob(v_ref.astype(np.float32), i_ref, vals, targets, 1.0)
    np.testing.assert_allclose(p.cpu().numpy(), p_ref.numpy(), rtol=2e-4, atol=1e-6)
    assert np.abs(rec.cpu().numpy() - rec_ref.numpy()).max() <= 1              # a near-tie at the k-th place may swap one neighbour


def test_knn_model_reads_faiss_index_file(dev, tmp_path):
    """`--index-file` pointing at a faiss `[OPQ,]IVF,PQ` file (what knn/index_builder.py writes): KNNModel reads it without
    faiss (faiss_io.read_ivfpq_index) and searches it on the device -- same neighbours as the index it was written from."""
    import json, os
    from gnnlm_amd import faiss_io
    from gnnlm_amd.ivfpq import IVFPQIndex
    from gnnlm_amd.knn_model import KNNModel
    rs = np.random.RandomState(14)
    N, d, V, M = 30000, 128, 40, 32
    centres = rs.randn(30, d).astype(np.float32)
    keys = (centres[rs.randint(0, 30, N)] + 0.5 * rs.randn(N, d)).astype(np.float16)
    vals = rs.randint(0, V, N).astype(np.int16)
    dd = tmp_path / "train_dstore"
    os.makedirs(dd)
    keys.tofile(dd / "keys.npy"); vals.tofile(dd / "vals.npy")
    json.dump({"dstore_size": N, "hidden_size": d, "vocab_size": V, "dstore_fp16": True, "val_size": 1}, open(dd / "info.json", "w"))
    built = IVFPQIndex.build(keys, 48, M, device=dev, cosine=True, nprobe=6, iters=5, seed=2)
    arrs = {a: getattr(built, a).cpu().numpy() for a in ("R", "coarse", "pq", "list_off", "list_ids", "list_codes")}
    f = str(dd / "faiss_store.cosine")
    faiss_io.write_ivfpq_index(f, arrs["R"], arrs["coarse"], arrs["pq"], arrs["list_off"], arrs["list_ids"], arrs["list_codes"], nprobe=1)
    m = KNNModel(f, str(dd), k=64, probe=6, no_load_keys=True, metric_type="do_not_recomp_ip", device=dev)
    assert isinstance(m.index, IVFPQIndex) and m.index.nprobe == 6 and m.index.packed_codes is not None and m.index.has_vals
    q = (centres[rs.randint(0, 30, 17)] + 0.5 * rs.randn(17, d)).astype(np.float32)
    qn = q / np.sqrt((q ** 2).sum(1, keepdims=True))
    v0, i0 = built.search(qn, 64)
    v1, i1 = m.index.search(qn, 64)
    assert np.array_equal(i0, i1) and np.array_equal(v0, v1)
    # an index whose coarse quantizer has another metric than the index is refused (and, without faiss or keys, by KNNModel)
    raw = bytearray(open(f, "rb").read())
    at = raw.index(b"IwPQ") + 4 + 4 + 8 + 8 + 8 + 1                            # the IVF index's metric_type (header: d, ntotal, 2 x dummy, is_trained)
    assert struct.unpack_from("<i", raw, at)[0] == 0
    struct.pack_into("<i", raw, at, 1)
    open(f, "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        IVFPQIndex.from_faiss_file(f, device=dev)
    with pytest.raises(ValueError):
        KNNModel(f, str(dd), k=64, probe=6, no_load_keys=True, metric_type="do_not_recomp_ip", device=dev)


def test_ivfpq_l2_index(dev, tmp_path):
    """The L2 metric (`IndexBuilder`'s default, knn/index_builder.py:26,118; `faiss_store.l2`): squared distances with residual
    codes, the nearest lists probed -- built here, searched on the device, against the float64 IVFADC oracle; the same index
    through a faiss-format file and `KNNModel(... metric_type="do_not_recomp_l2")`, and through `run_index_build --metric l2`."""
    import json, os
    from gnnlm_amd import faiss_io, run_index_build
    from gnnlm_amd.ivfpq import IVFPQIndex
    from gnnlm_amd.knn_model import KNNModel
    from oracle import ivfpq as oivf
    rs = np.random.RandomState(31)
    N, d, M, nlist, V = 40_000, 128, 32, 40, 50
    centres = 2.0 * rs.randn(25, d).astype(np.float32)
    keys = (centres[rs.randint(0, 25, N)] + 0.7 * rs.randn(N, d)).astype(np.float32)
    q = (centres[rs.randint(0, 25, 33)] + 0.7 * rs.randn(33, d)).astype(np.float32)
    idx = IVFPQIndex.build(keys, nlist, M, device=dev, cosine=False, metric="l2", nprobe=8, iters=6, seed=4)
    assert idx.metric == "l2" and idx.tiles is None and idx.packed_codes is None and idx.list_term is not None
    arrs = [getattr(idx, a).cpu().numpy() for a in ("R", "coarse", "pq", "list_off", "list_ids", "list_codes")]
    exact = np.argsort(((q[:, None, :].astype(np.float64) - keys[None].astype(np.float64)) ** 2).sum(-1), axis=1)[:, :10]
    for k in (1024, 64):
        dist, ids = idx.search(q, k)
        d_ref, i_ref = oivf.search(q, *arrs, k=k, nprobe=8, metric="l2")
        assert (np.diff(dist, axis=1) >= 0).all()                            # ascending squared distances
        assert np.mean([len(set(a) & set(b)) / k for a, b in zip(ids, i_ref)]) > 0.998
        np.testing.assert_allclose(dist, d_ref, rtol=2e-4, atol=2e-4)
    assert np.mean([len(set(a[:50]) & set(b)) / 10 for a, b in zip(i_ref, exact)]) > 0.8   # (and the index finds the true neighbours)
    # fewer keys than k in the probed lists: +inf / -1 padding
    small = IVFPQIndex(*[getattr(idx, a) for a in ("R", "coarse", "pq", "list_off", "list_ids", "list_codes")], nprobe=1, cosine=False, metric="l2")
    dist, ids = small.search(q[:4], 2000)
    assert (ids[:, -1] == -1).all() and np.isinf(dist[:, -1]).all() and (dist[:, -1] > 0).all()
    # the faiss file of such an index -> KNNModel
    vals = rs.randint(0, V, N).astype(np.int16)
    dd = tmp_path / "train_dstore"
    os.makedirs(dd)
    keys.astype(np.float16).tofile(dd / "keys.npy"); vals.tofile(dd / "vals.npy")
    json.dump({"dstore_size": N, "hidden_size": d, "vocab_size": V, "dstore_fp16": True, "val_size": 1}, open(dd / "info.json", "w"))
    f = str(dd / "faiss_store.l2")
    faiss_io.write_ivfpq_index(f, *arrs, nprobe=1, metric="l2")
    m = KNNModel(f, str(dd), k=64, probe=8, no_load_keys=True, metric_type="do_not_recomp_l2", device=dev)
    assert isinstance(m.index, IVFPQIndex) and m.index.metric == "l2" and not m.cosine
    sims, knns = m.search_sims(torch.from_numpy(q).to(dev), 64)
    d_ref, i_ref = oivf.search(q, *arrs, k=64, nprobe=8, metric="l2")
    assert np.mean([len(set(a) & set(b)) / 64 for a, b in zip(knns.cpu().numpy(), i_ref)]) > 0.998
    np.testing.assert_allclose(sims.cpu().numpy(), -d_ref, rtol=2e-4, atol=2e-4)   # knn_model.py:139: sims = -dists
    targets = torch.from_numpy(vals[i_ref[:, 1]].astype(np.int64)).to(dev)
    p, rec = m.get_knn_prob(torch.from_numpy(q).to(dev), targets=targets, t=10.0, return_recall=True)
    p_ref, rec_ref = oknn.knn_target_prob((-d_ref).astype(np.float32), i_ref, vals, targets.cpu().numpy(), 10.0)
    np.testing.assert_allclose(p.cpu().numpy(), p_ref.numpy(), rtol=2e-3, atol=1e-6)
    assert np.abs(rec.cpu().numpy() - rec_ref.numpy()).max() <= 1
    # the producer
    os.remove(f)
    out = run_index_build.main(run_index_build.get_parser().parse_args(
        ["--dstore-dir", str(dd), "--index-type", "OPQ32_128,IVF40,PQ32", "--metric", "l2", "--nprobe", "8", "--opq-iters", "2"]))
    assert out.endswith("faiss_store.l2.gnnlm.npz")
    m2 = KNNModel(str(dd / "faiss_store.l2"), str(dd), k=64, probe=8, no_load_keys=True, metric_type="do_not_recomp_l2", device=dev)
    assert m2.index.metric == "l2"
    d2, i2 = m2.index.search(q, 10)
    # (0.52-0.54 over 12 builds -- k-means on the device accumulates with float atomics, so the produced index varies a little from run to
    # run; the line checks that the producer's index finds true neighbours at all, not the quantiser's quality: 0.5 flaked once in ~20 runs)
    assert np.mean([len(set(a) & set(b)) / 10 for a, b in zip(i2, exact)]) > 0.4


def test_opq_training_lowers_the_quantisation_error(dev):
    """IVFPQIndex.build(opq_iters > 0): the trained rotation (train_opq: PQ rounds + orthogonal Procrustes, the reference's
    `OPQ64_1024` block) stays orthonormal and loses less than the random rotation it starts from -- reconstruction error of
    the added keys and recall of the true nearest neighbours -- on keys with a decaying spectrum and correlated dimensions."""
    from gnnlm_amd.ivfpq import IVFPQIndex
    rs = np.random.RandomState(21)
    N, d, M, nlist, k = 60_000, 128, 16, 16, 32
    spectrum = 0.97 ** np.arange(d)
    Q = np.linalg.qr(rs.randn(d, d))[0]
    keys = ((rs.randn(N, d) * spectrum) @ Q).astype(np.float32)
    q = ((rs.randn(64, d) * spectrum) @ Q).astype(np.float32)
    exact = np.argsort(-(q.astype(np.float64) @ keys.T.astype(np.float64)), axis=1)[:, :k]
    res = {}
    for name, it in (("random", 0), ("opq", 8)):
        idx = IVFPQIndex.build(keys, nlist, M, device=dev, cosine=False, nprobe=nlist, iters=6, seed=5, opq_iters=it)
        R = idx.R.cpu().numpy().astype(np.float64)
        assert np.abs(R @ R.T - np.eye(d)).max() < 1e-4
        off, codes, ids = idx.list_off.cpu().numpy(), idx.list_codes.cpu().numpy().astype(np.int64), idx.list_ids.cpu().numpy()
        lists = np.searchsorted(off, np.arange(N), side="right") - 1
        rec = idx.coarse.cpu().numpy()[lists] + idx.pq.cpu().numpy()[np.arange(M)[None, :], codes].reshape(N, d)
        err = float((((keys[ids].astype(np.float64) @ R.T) - rec) ** 2).sum(1).mean())
        _, i = idx.search(q, k)
        res[name] = (err, np.mean([len(set(a) & set(b)) / k for a, b in zip(i, exact)]))
    assert res["opq"][0] < 0.9 * res["random"][0], res
    assert res["opq"][1] >= res["random"][1] - 0.01, res


@pytest.mark.parametrize("metric", ["cosine", "l2"])
def test_knn_model_reads_faiss_flat_file(dev, tmp_path, metric):
    """`--index-file` pointing at an `IDMap,,Flat` file (index_builder.py:49-53: the auto type for small datastores; the keys
    are normalised before they are added for "cosine", :90-95): exact search over the FILE's vectors with the file's ids."""
    import json, os
    from gnnlm_amd import faiss_io
    from gnnlm_amd.knn_model import ExactIndex, KNNModel
    rs = np.random.RandomState(15)
    N, d, V, k, n = 5000, 64, 40, 32, 19
    keys = rs.randn(N, d).astype(np.float32)
    vals = rs.randint(0, V, N).astype(np.int16)
    dd = tmp_path / "train_dstore"
    os.makedirs(dd)
    keys.astype(np.float16).tofile(dd / "keys.npy"); vals.tofile(dd / "vals.npy")
    json.dump({"dstore_size": N, "hidden_size": d, "vocab_size": V, "dstore_fp16": True, "val_size": 1}, open(dd / "info.json", "w"))
    ids = rs.permutation(N).astype(np.int64)                             # add_with_ids in some other order than the rows
    xb = keys[ids] / np.sqrt((keys[ids] ** 2).sum(1, keepdims=True)) if metric == "cosine" else keys[ids]
    f = str(dd / f"faiss_store.{metric}")
    faiss_io.write_flat_index(f, xb, ids, metric="ip" if metric == "cosine" else "l2")
    fn = "do_not_recomp_ip" if metric == "cosine" else "do_not_recomp_l2"
    m = KNNModel(f, str(dd), k=k, no_load_keys=True, metric_type=fn, device=dev)
    assert isinstance(m.index, ExactIndex) and m.index.ntotal == N
    q = rs.randn(n, d).astype(np.float32)
    sims, knns = m.search_sims(torch.from_numpy(q).to(dev), k)
    sims, knns = sims.cpu().numpy(), knns.cpu().numpy()
    if metric == "cosine":
        qn = q / np.sqrt((q ** 2).sum(1, keepdims=True))
        ref = qn.astype(np.float64) @ xb.T.astype(np.float64)
    else:
        ref = -((q[:, None, :].astype(np.float64) - xb[None].astype(np.float64)) ** 2).sum(-1)
    order = np.argsort(-ref, axis=1, kind="stable")[:, :k]
    assert np.array_equal(knns, ids[order])
    np.testing.assert_allclose(sims, np.take_along_axis(ref, order, 1), rtol=2e-5, atol=2e-5)
    targets = torch.from_numpy(vals[ids[order[:, 3]]].astype(np.int64)).to(dev)
    p, rec = m.get_knn_prob(torch.from_numpy(q).to(dev), targets=targets, t=1.0, return_recall=True)
    p_ref, rec_ref = oknn.knn_target_prob(np.take_along_axis(ref, order, 1).astype(np.float32), ids[order], vals, targets.cpu().numpy(), 1.0)
    np.testing.assert_allclose(p.cpu().numpy(), p_ref.numpy(), rtol=2e-4, atol=1e-6)
    assert np.array_equal(rec.cpu().numpy(), rec_ref.numpy())


@pytest.mark.parametrize("k", [64, 1024, 2000])
@pytest.mark.parametrize("largest", [True, False])
def test_topk_select_from_ragged_candidate_lists(dev, k, largest):
    """init + per-row ids + row_ncols (what the exact re-score of an IVF-PQ search hands over: one ragged candidate list per
    query, a few times k long -- the counting pre-pass applies from 2 KP columns on): exact ids with ties by ascending id,
    empty and short rows, negative (skipped) ids."""
    from gnnlm_amd import ops
    rs = np.random.RandomState(3 * k + largest)
    n, W = 11, 20000
    scores = rs.randn(n, W).astype(np.float32)
    scores[:, ::30] = scores[:, 1::30]                                    # exact ties
    scores[1] = np.round(scores[1] * 3) / 3                               # thousands of ties
    ids = np.stack([rs.permutation(10_000_000_000 + np.arange(W)) for _ in range(n)]).astype(np.int64)
    ids[2, ::9] = -1                                                      # skipped entries
    ncols = np.array([W, W, W, 0, 1, k - 1, k, k + 1, 4096, 8192, 8193], dtype=np.int32)
    args = dict(largest=largest, init=True, row_ncols=torch.from_numpy(ncols).to(dev), ids=torch.from_numpy(ids).to(dev))
    s_dev = torch.from_numpy(scores).to(dev)
    bv = torch.empty(n, k, device=dev)
    bi = torch.empty(n, k, device=dev, dtype=torch.int64)
    ops.topk_merge(s_dev, bv, bi, **args)
    bv, bi = bv.cpu().numpy(), bi.cpu().numpy()
    for r in range(n):
        ok = (np.arange(W) < ncols[r]) & (ids[r] >= 0)
        v, i = ref_topk(scores[r][ok], ids[r][ok], k, largest)
        m = min(k, int(ok.sum()))
        assert np.array_equal(bi[r, :m], i[:m]), r
        np.testing.assert_array_equal(bv[r, :m], v[:m])
        assert (bi[r, m:] == -1).all()


def test_ivfpq_one_billion_index_family(dev):
    """The index family of the One Billion Word recipe (`OPQ16_64,IVF1048576,PQ16`, nprobe 32, k = 256:
    gnnlm_scripts/one_billion/find_knn.sh:9,20): a RECTANGULAR OPQ matrix (the rotation also reduces 1024 -> 64 dimensions),
    16 sub-quantizers of 4 dimensions, far more lists than a query probes and short lists (here 8192 lists over 400 k keys:
    ~49 keys each, many empty) -- against the float64 IVFADC oracle over the same arrays."""
    from gnnlm_amd.ivfpq import IVFPQIndex
    from oracle import ivfpq as oivf
    rs = np.random.RandomState(31)
    N, d_in, d_out, nlist, M, nprobe, k = 400_000, 256, 64, 8192, 16, 32, 256
    centres = rs.randn(300, d_in).astype(np.float32)
    keys = (centres[rs.randint(0, 300, N)] + 0.5 * rs.randn(N, d_in).astype(np.float32)).astype(np.float32)
    P = np.linalg.qr(rs.randn(d_in, d_out))[0].T.astype(np.float32)          # [d_out, d_in] with orthonormal rows: the OPQ projection
    inner = IVFPQIndex.build(keys @ P.T, nlist, M, device=dev, cosine=False, nprobe=nprobe, iters=4, seed=2)
    R = (inner.R.cpu().numpy() @ P).astype(np.float32)                       # rotation of the projected space after the projection
    index = IVFPQIndex(torch.from_numpy(R).to(dev), inner.coarse, inner.pq, inner.list_off, inner.list_ids, inner.list_codes,
                       nprobe=nprobe, cosine=False)
    assert index.R.shape == (d_out, d_in) and index.M == 16 and index.dsub == 4 and index.tiles is None
    lens = (index.list_off[1:] - index.list_off[:-1]).cpu().numpy()
    assert (lens == 0).any() and lens.max() < 4000
    q = (centres[rs.randint(0, 300, 50)] + 0.5 * rs.randn(50, d_in)).astype(np.float32)
    v, i = index.search(q, k)
    arrs = [R, inner.coarse.cpu().numpy(), inner.pq.cpu().numpy(), inner.list_off.cpu().numpy(), inner.list_ids.cpu().numpy(),
            inner.list_codes.cpu().numpy()]
    v_ref, i_ref = oivf.search(q, *arrs, k=k, nprobe=nprobe)
    ok = i_ref >= 0
    assert np.array_equal(i >= 0, ok)                                        # the same number of neighbours found (-1 padding beyond)
    same = np.mean([len(set(a[a >= 0]) & set(b[b >= 0])) / max(1, (b >= 0).sum()) for a, b in zip(i, i_ref)])
    assert same > 0.998, same
    np.testing.assert_allclose(v[ok], v_ref[ok], rtol=2e-5, atol=2e-4)


def test_topk_select_duplicate_ids_among_ties(dev):
    """Caller-supplied ids that repeat inside a run of values tied at the cut (the select kernel's id cut then admits more than
    it needs): no strictly better entry may be lost, the row stays sorted, the tail is made of tied entries."""
    from gnnlm_amd import ops
    n, ncols, k = 6, 3000, 256
    rs = np.random.RandomState(17)
    sc = rs.randn(n, ncols).astype(np.float32)
    ids = np.stack([rs.permutation(10 * ncols)[:ncols] for _ in range(n)]).astype(np.int64)
    for r in range(n):
        order = np.argsort(-sc[r])
        kth = sc[r, order[k - 40]]
        tied = order[k - 40:k + 400]                        # 440 entries tied at the cut value ...
        sc[r, tied] = kth
        ids[r, tied] = ids[r, tied[0]] if r % 2 else ids[r, tied] % 7      # ... that share one id / seven ids
    bv = torch.empty(n, k, device=dev)
    bi = torch.empty(n, k, device=dev, dtype=torch.int64)
    ops.topk_merge(torch.from_numpy(sc).to(dev), bv, bi, ids=torch.from_numpy(ids).to(dev), largest=True, init=True)
    bv, bi = bv.cpu().numpy(), bi.cpu().numpy()
    for r in range(n):
        want = np.sort(sc[r])[::-1][:k]
        assert np.array_equal(bv[r], want), r              # every strictly better value present, tail = the tied value
        better = sc[r] > want[-1]
        assert set(ids[r, better]) <= set(bi[r]), r
