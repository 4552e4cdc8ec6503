"""Host-side mirrors of the reference interface, CPU parts (no kernel launches):
DataStore / path layout / TorchPQCodec tables+encode+compute_sim / driver helpers, against the golden
vectors produced by the reference's own code."""
import json
import os

import struct

import numpy as np
import pytest
import torch

from gnnlm_amd import path_utils
from gnnlm_amd.data_store import DataStore
from gnnlm_amd.eval_lm import block_ranges, get_parser
from gnnlm_amd.pq_wrapper import TorchPQCodec


@pytest.mark.parametrize("name", ["fp16_i16", "fp16_i32", "fp32_i32", "fp16_v2"])
def test_datastore_mirror(golden, tmp_path, name):
    g = golden("datastore")
    info = json.loads(bytes(g[name + ".info"]).decode())
    d = tmp_path / "x_dstore"
    d.mkdir()
    (d / "keys.npy").write_bytes(bytes(g[name + ".keys_raw"]))
    (d / "vals.npy").write_bytes(bytes(g[name + ".vals_raw"]))
    json.dump(info, open(d / "info.json", "w"))
    for use_memory in (False, True):
        ds = DataStore.from_pretrained(str(d), use_memory=use_memory)
        assert ds.info == info
        assert np.array_equal(np.array(ds.keys), g[name + ".keys"]) and ds.keys.dtype == g[name + ".keys"].dtype
        assert np.array_equal(np.array(ds.vals), g[name + ".vals"]) and ds.vals.dtype == g[name + ".vals"].dtype
    ds = DataStore.from_pretrained(str(d), no_load_keys=True)
    assert not hasattr(ds, "keys")
    assert DataStore.exists(str(d))
    with pytest.raises(FileNotFoundError):
        DataStore.from_pretrained(str(tmp_path / "missing"))


def test_path_layout():
    assert path_utils.feature_path("D", "test") == os.path.join("D", "test_dstore", "keys.npy")
    assert path_utils.value_path("D", "train") == os.path.join("D", "train_dstore", "vals.npy")
    assert path_utils.quantized_feature_path("D", "train") == os.path.join("D", "train_dstore", "quantized-keys.npy")
    assert path_utils.neighbor_path("D", "valid", 128) == os.path.join("D", "valid_dstore", "neighbors.mmap.128")
    assert path_utils.quantizer_path("D") == os.path.join("D", "quantizer")
    assert path_utils.quantizer_path("D", "-x", True) == os.path.join("D", "quantizer-norm-x")
    assert path_utils.dstore_path("D", "test") == os.path.join("D", "test_dstore")


@pytest.mark.parametrize("case", ["sq_pre", "sq_pre_nob", "rect_pre", "nopre"])
def test_codec_tables_encode_sim(golden, case):
    g = golden("pq")
    A = g[f"{case}.A"] if f"{case}.A" in g else None
    b = g[f"{case}.b"] if f"{case}.b" in g else None
    for metric in ("ip", "l2"):
        c = TorchPQCodec.from_arrays(g[f"{case}.cen"], A, b, metric)
        codes = torch.from_numpy(g[f"{case}.codes"].copy())                           # encode itself: GPU test
        with pytest.raises(Exception):
            c.encode(torch.from_numpy(g[f"{case}.x"].copy()))                         # host tensor: no CPU fallback
        np.testing.assert_allclose(c.norm2_centroids_torch.numpy(), g[f"{case}.norm2"], rtol=1e-6)
        np.testing.assert_allclose(c.sdc_table_torch.numpy()[:, :8, :8], g[f"{case}.{metric}.sdc_corner"], atol=2e-6)
        np.testing.assert_allclose(c.compute_sim(codes[:5], codes).numpy(), g[f"{case}.{metric}.sim"], rtol=1e-5, atol=1e-4)
    names = set(dict(c.named_buffers()))
    assert {"centroids_torch", "norm2_centroids_torch", "sdc_table_torch"} <= names
    assert ("A" in names) == (A is not None)
    with pytest.raises(Exception):
        c.decode(torch.zeros(2, c.centroids_torch.shape[0], dtype=torch.uint8))       # host tensor: no CPU fallback


def test_codec_save_load(tmp_path, golden):
    g = golden("pq")
    c = TorchPQCodec.from_arrays(g["sq_pre.cen"], g["sq_pre.A"], g["sq_pre.b"])
    c.save(str(tmp_path / "q.npz"))
    c2 = TorchPQCodec.from_file(str(tmp_path / "q.npz"))
    for (n1, b1), (n2, b2) in zip(c.named_buffers(), c2.named_buffers()):
        assert n1 == n2 and torch.equal(b1, b2)


def test_block_ranges_and_flags():
    assert block_ranges(10, 4) == [(0, 0, 4), (4, 4, 8), (8, 8, 10)]
    assert block_ranges(10, 4, 3) == [(0, 0, 4), (1, 4, 8), (5, 8, 10)]
    assert block_ranges(0, 4) == []
    a = get_parser().parse_args(["DATA", "--path", "m.pt"])
    # defaults of the reference flags (language_modeling.py:97-153, options.py:472-501)
    assert (a.gcn_k, a.k, a.lmbda, a.knn_sim_func, a.neighbor_context, a.temperature, a.gcn_context_window,
            a.invalid_neighbor_context, a.probe) == (1024, 1024, 0.0, "do_not_recomp_ip", "(2, 2)", 1.0, 0, 1536, 8)
    r = get_parser().parse_args("DATA --path m.pt --graph --neighbor-context 2 --gcn-k 128 --use-precompute-feat "
                                "--sample-break-mode none --max-tokens 256 --tokens-per-sample 256 --softmax-batch 3072 "
                                "--gcn-context-window 0 --gen-subset test --knn-keytype keytype --knnlm --k 1024 "
                                "--lmbda 0.25 --dstore-dir D/train_dstore --index-file D/faiss_store.cosine "
                                "--temperature 0.01 --knn-sim-func do_not_recomp_ip".split() +
                                ["--model-overrides", "{'orig_prob_ratio': 0.0, 'max_target_positions': 256, 'add_bias': False}"])
    assert r.graph and r.knnlm and r.gcn_k == 128 and r.lmbda == 0.25


def test_faiss_quantizer_file_round_trip(tmp_path):
    """The `quantizer` file layout (faiss IndexPreTransform(OPQ) -> IndexPQ, quantize_features.py:108-109) restated in
    faiss_io.py: writer and reader agree, with and without the pre-transform / bias, and garbage is refused."""
    import pytest
    from gnnlm_amd.faiss_io import read_pq_quantizer, write_pq_quantizer
    rs = np.random.RandomState(0)
    cen = rs.randn(8, 256, 4).astype(np.float32)
    A = rs.randn(32, 48).astype(np.float32)
    b = rs.randn(32).astype(np.float32)
    for a_, b_ in [(A, b), (A, None), (None, None)]:
        f = str(tmp_path / "quantizer")
        write_pq_quantizer(f, cen, a_, b_)
        q = read_pq_quantizer(f)
        assert np.array_equal(q["centroids"], cen) and q["metric"] == "ip"
        assert (q["A"] is None) == (a_ is None) and (a_ is None or np.array_equal(q["A"], a_))
        assert b_ is None or np.array_equal(q["b"], b_)
    raw = open(f, "rb").read()
    assert raw[:4] == b"IxPq" and struct.unpack_from("<i", raw, 4)[0] == 32                  # header: fourcc, int32 d
    open(f, "wb").write(b"IxHN" + raw[4:])
    with pytest.raises(ValueError):
        read_pq_quantizer(f)


def test_faiss_flat_index_file_round_trip(tmp_path):
    """`IDMap,,Flat` / `Flat` files (the auto index type of knn/index_builder.py:49-53 for small datastores): layout spot checks,
    writer and reader agree, both metrics, other kinds are refused."""
    import pytest
    from gnnlm_amd.faiss_io import read_flat_index, sniff, write_flat_index
    rs = np.random.RandomState(2)
    xb = rs.randn(37, 12).astype(np.float32)
    ids = rs.permutation(1000)[:37].astype(np.int64)
    f = str(tmp_path / "faiss_store.l2")
    for metric in ("ip", "l2"):
        for ids_ in (ids, None):
            write_flat_index(f, xb, ids_, metric=metric)
            assert sniff(f) == ("IxMp" if ids_ is not None else ("IxFI" if metric == "ip" else "IxF2"))
            z = read_flat_index(f)
            assert np.array_equal(z["xb"], xb) and z["metric"] == metric
            assert (z["ids"] is None) == (ids_ is None) and (ids_ is None or np.array_equal(z["ids"], ids_))
    write_flat_index(f, xb, ids, metric="l2")
    raw = open(f, "rb").read()
    # IxMp | header | IxF2 | header (int32 d, int64 ntotal, ..., int32 metric = 1) | uint64 n floats | data | uint64 n ids | ids
    assert raw[:4] == b"IxMp" and raw[37:41] == b"IxF2" and struct.unpack_from("<iq", raw, 41) == (12, 37)
    assert struct.unpack_from("<i", raw, 41 + 29)[0] == 1 and struct.unpack_from("<Q", raw, 74)[0] == 37 * 12
    assert struct.unpack_from("<Q", raw, 82 + 37 * 12 * 4)[0] == 37
    open(f, "wb").write(b"IxPq" + raw[4:])
    with pytest.raises(ValueError):
        read_flat_index(f)


def test_faiss_ivfpq_index_file_round_trip(tmp_path):
    """The kNN index file layout (faiss IndexPreTransform(OPQ) -> IndexIVFPQ over an IndexFlatIP, index_builder.py:79-150)
    restated in faiss_io.py: byte-level spot checks of the layout, writer and reader agree (dense and sparse list-size
    tables, with and without the OPQ matrix), other index kinds are refused."""
    import pytest
    from gnnlm_amd.faiss_io import read_ivfpq_index, sniff, write_ivfpq_index
    rs = np.random.RandomState(1)
    d, M, nlist = 32, 8, 6
    R = rs.randn(d, d).astype(np.float32)
    coarse = rs.randn(nlist, d).astype(np.float32)
    pq = rs.randn(M, 256, d // M).astype(np.float32)
    f = str(tmp_path / "faiss_store.cosine")
    for sizes in ([5, 0, 70, 1, 3, 64], [0, 0, 9, 0, 0, 0]):                                   # "full" and "sprs" size tables
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        N = int(off[-1])
        ids = rs.permutation(N).astype(np.int64)
        codes = rs.randint(0, 256, (N, M)).astype(np.uint8)
        for R_ in (R, None):
            write_ivfpq_index(f, R_, coarse, pq, off, ids, codes, nprobe=3)
            assert sniff(f) == ("IxPT" if R_ is not None else "IwPQ")
            z = read_ivfpq_index(f)
            assert (z["R"] is None) == (R_ is None) and (R_ is None or np.array_equal(z["R"], R_))
            assert np.array_equal(z["coarse"], coarse) and np.array_equal(z["pq"], pq)
            assert np.array_equal(z["list_off"], off) and np.array_equal(z["list_ids"], ids) and np.array_equal(z["list_codes"], codes)
            assert z["nprobe"] == 3 and z["metric"] == "ip" and z["coarse_metric"] == "ip" and z["by_residual"]
    # `IVF{n}_HNSW32` (what index_builder.py:60-64 picks by itself for >= 10^6 keys): the coarse quantizer is an IndexHNSWFlat;
    # the reader takes the flat storage behind the graph
    write_ivfpq_index(f, R, coarse, pq, off, ids, codes, nprobe=5, coarse_kind="hnsw")
    z = read_ivfpq_index(f)
    assert z["coarse_kind"] == "hnsw" and np.array_equal(z["coarse"], coarse) and np.array_equal(z["list_codes"], codes) and z["nprobe"] == 5
    assert b"IHNf" in open(f, "rb").read()
    write_ivfpq_index(f, None, coarse, pq, off, ids, codes, nprobe=3)
    raw = open(f, "rb").read()
    # IwPQ | header (int32 d, int64 ntotal, 2 x int64, bool, int32 metric) | uint64 nlist | uint64 nprobe | IxFI ...
    assert raw[:4] == b"IwPQ" and struct.unpack_from("<iq", raw, 4) == (d, 9)
    assert struct.unpack_from("<QQ", raw, 4 + 33) == (nlist, 3) and raw[4 + 33 + 16:4 + 33 + 20] == b"IxFI"
    assert b"ilar" in raw and b"sprs" in raw
    open(f, "wb").write(b"IxHN" + raw[4:])
    with pytest.raises(ValueError):
        read_ivfpq_index(f)


def test_ivfpq_group_table():
    """The task table of the int8-MFMA IVF-PQ scan (gnnlm_ivfpq_scan8, include/gnnlm.h): (query, probe) pairs -> groups of at
    most 8 queries that probe the same list, sorted by list; every pair exactly once, full groups except the last of a list,
    -1 probes dropped, the output offsets of the threshold pass -- pure torch, device-agnostic (here on the CPU)."""
    import torch
    from gnnlm_amd.ivfpq import build_groups
    rs = np.random.RandomState(5)
    nlist, nq, P, seg = 24, 61, 7, 1024
    pl = torch.from_numpy(np.stack([rs.permutation(nlist)[:P] for _ in range(nq)]))
    pl[3, 2] = -1
    pl[11, :] = -1
    grp_list, grp_q, n_groups, G, grp_out = build_groups(pl, nlist, seg=seg)
    ng = int(n_groups.item())
    assert G == nq * P // 8 + nlist + 1 and ng <= G
    grp_list, grp_q, grp_out = grp_list.numpy(), grp_q.numpy(), grp_out.numpy()
    assert (grp_list[ng:] == -1).all() and (grp_q[ng:] == -1).all() and (grp_out[ng:] == -1).all()
    live = grp_list[:ng]
    assert (np.diff(live[live >= 0]) >= 0).all()                             # sorted by list (the no-list bucket comes last)
    got = sorted((int(l), int(q)) for l, row in zip(grp_list[:ng], grp_q[:ng]) for q in row if q >= 0 and l >= 0)
    want = sorted((int(l), r) for r, row in enumerate(pl.numpy()) for l in row if l >= 0)
    assert got == want                                                       # every pair exactly once
    for l in np.unique(live[live >= 0]):
        rows = grp_q[:ng][live == l]
        assert ((rows >= 0).sum(1)[:-1] == 8).all() and (rows >= 0).sum(1)[-1] >= 1
    # the segment of pair (q, slot): (q * P + slot) * seg, for the pairs that name a list
    pln = pl.numpy()
    for g in range(ng):
        for j in range(8):
            q, o = grp_q[g, j], grp_out[g, j]
            if q < 0 or grp_list[g] < 0:
                assert o == -1 or grp_list[g] < 0
                continue
            slot = int(np.nonzero(pln[q] == grp_list[g])[0][0])
            assert o == (q * P + slot) * seg
    # without seg: the same table
    a = build_groups(pl, nlist)
    assert np.array_equal(a[0].numpy(), grp_list) and np.array_equal(a[1].numpy(), grp_q) and int(a[2].item()) == ng


def test_unsupported_faiss_index_files_are_refused(tmp_path):
    """IVFPQIndex.from_faiss_file covers what knn/index_builder.py builds (residual codes, the coarse quantizer in the index's
    metric); files outside that family are refused BEFORE anything touches the device: ``by_residual = False`` and a coarse
    quantizer of the other metric."""
    from gnnlm_amd.faiss_io import read_ivfpq_index, write_ivfpq_index
    from gnnlm_amd.ivfpq import IVFPQIndex
    rs = np.random.RandomState(2)
    d, M, nlist = 32, 8, 4
    coarse, pq = rs.randn(nlist, d).astype(np.float32), rs.randn(M, 256, d // M).astype(np.float32)
    off = np.array([0, 3, 3, 10, 12], np.int64)
    ids, codes = np.arange(12, dtype=np.int64), rs.randint(0, 256, (12, M)).astype(np.uint8)
    f = str(tmp_path / "faiss_store.cosine")
    write_ivfpq_index(f, None, coarse, pq, off, ids, codes, nprobe=2)
    raw = bytearray(open(f, "rb").read())
    at = raw.index(b"IxFI")                                     # the coarse quantizer: IndexFlatIP -> IndexFlatL2 (metric field too)
    assert read_ivfpq_index(f)["by_residual"]
    bad = bytearray(raw)
    bad[at:at + 4] = b"IxF2"
    struct.pack_into("<i", bad, at + 4 + 29, 1)                 # header: int32 d, int64 x 3, bool, int32 metric
    open(f, "wb").write(bytes(bad))
    z = read_ivfpq_index(f)
    assert z["metric"] == "ip" and z["coarse_metric"] == "l2"
    with pytest.raises(ValueError, match="residual codes"):
        IVFPQIndex.from_faiss_file(f, device="cuda")            # raises while still on the host
    # by_residual = False: the bool in front of code_size, right behind the direct map (int8 type + empty vector)
    flat_end = at + 4 + 33 + 8 + 4 * nlist * d
    bad = bytearray(raw)
    assert bad[flat_end] == 0 and struct.unpack_from("<Q", bad, flat_end + 1) == (0,) and bad[flat_end + 9] == 1
    bad[flat_end + 9] = 0
    open(f, "wb").write(bytes(bad))
    assert not read_ivfpq_index(f)["by_residual"]
    with pytest.raises(ValueError, match="residual codes"):
        IVFPQIndex.from_faiss_file(f, device="cuda")


def _write_split(tmp_path, n_tok=50, d=8, kg=4, vocab=100, fp16=True):
    data = tmp_path / "data-bin"
    rs = np.random.RandomState(0)
    for split, n in (("train", 200), ("test", n_tok)):
        p = data / f"{split}_dstore"
        os.makedirs(p, exist_ok=True)
        rs.randn(n, d).astype(np.float16).tofile(p / "keys.npy")
        (4 + rs.randint(0, vocab - 4, n)).astype(np.int16).tofile(p / "vals.npy")
        json.dump({"dstore_size": n, "hidden_size": d, "vocab_size": vocab, "dstore_fp16": fp16, "val_size": 1}, open(p / "info.json", "w"))
    rs.randint(0, 200, (n_tok, kg)).astype(np.int64).tofile(data / "test_dstore" / f"neighbors.mmap.{kg}")
    return data


def _write_fairseq_bin(data, split, tokens, sizes, dtype=np.uint16):
    code = {np.uint8: 1, np.int16: 3, np.int32: 4, np.int64: 5, np.uint16: 8}[dtype]
    with open(data / f"{split}.idx", "wb") as f:                # MMapIndexedDataset.Index.writer (indexed_dataset.py:354-390)
        f.write(b"MMIDIDX\x00\x00" + struct.pack("<Q", 1) + struct.pack("<B", code) + struct.pack("<Q", len(sizes)))
        f.write(np.asarray(sizes, np.int32).tobytes())
        f.write((np.concatenate([[0], np.cumsum(sizes)[:-1]]) * np.dtype(dtype).itemsize).astype(np.int64).tobytes())
    np.asarray(tokens, dtype).tofile(data / f"{split}.bin")


def test_driver_input_guards(tmp_path):
    """eval_lm.check_tables: the per-token files of a split must describe the same tokens (sizes against info.json), and when the
    binarised split is there, vals.npy must BE its token stream (the reference takes targets from fairseq's dataset; the driver
    takes them from the datastore)."""
    from gnnlm_amd.eval_lm import check_tables, fairseq_token_stream
    data = _write_split(tmp_path)
    args = get_parser().parse_args([str(data), "--path", "x", "--gen-subset", "test", "--gcn-k", "4"])
    info = json.load(open(data / "test_dstore" / "info.json"))
    check_tables(args, info, 200)                               # consistent, no .bin: passes
    assert fairseq_token_stream(str(data), "test") is None
    vals = np.fromfile(data / "test_dstore" / "vals.npy", dtype=np.int16)
    _write_fairseq_bin(data, "test", vals, [20, 17, 13])
    assert np.array_equal(fairseq_token_stream(str(data), "test"), vals)
    check_tables(args, info, 200)
    _write_fairseq_bin(data, "test", np.concatenate([vals, [5, 6, 7]]), [20, 17, 16])     # a longer stream (--first): the prefix counts
    check_tables(args, info, 200)
    wrong = vals.copy()
    wrong[31] += 1
    _write_fairseq_bin(data, "test", wrong, [50], dtype=np.int32)
    with pytest.raises(ValueError, match="row 31"):
        check_tables(args, info, 200)
    _write_fairseq_bin(data, "test", vals[:40], [40])
    with pytest.raises(ValueError, match="holds 40 tokens"):
        check_tables(args, info, 200)
    os.remove(data / "test.bin")
    os.remove(data / "test.idx")
    # neighbours written for another k / another split length
    nb = data / "test_dstore" / "neighbors.mmap.4"
    raw = open(nb, "rb").read()
    open(nb, "wb").write(raw[:-32])
    with pytest.raises(ValueError, match="do not describe the same tokens"):
        check_tables(args, info, 200)
    open(nb, "wb").write(raw)
    open(data / "test_dstore" / "vals.npy", "ab").write(b"\0\0")
    with pytest.raises(ValueError, match="vals.npy"):
        check_tables(args, info, 200)


def test_quantize_features_sample_and_index_string():
    """The training sample of knn/quantize_features.py:79-90 (first rows of 100 equal parts, remainder in the last) and the index
    strings of the recipes (find_knn.sh:32-38 wiki103, one_billion/find_knn.sh:24)."""
    from gnnlm_amd.quantize_features import parse_index, training_sample
    keys = np.arange(1000 * 3, dtype=np.float32).reshape(1000, 3)
    s = training_sample(keys, 1000, 250)                        # 2 rows per part, the last part asks for 250 - 2 * 99 = 52 of its 10
    want = [p * 10 + i for p in range(99) for i in range(2)] + [990 + i for i in range(10)]
    assert np.array_equal(s[:, 0] / 3, np.array(want, np.float32))
    s = training_sample(keys, 1000, 300)                        # the recipe's case: chunk_size % 100 == 0, no remainder
    assert np.array_equal(s[:, 0] / 3, np.array([p * 10 + i for p in range(100) for i in range(3)], np.float32))
    s = training_sample(keys, 1000, 5000)                       # chunk_size > store: every part whole = the whole table
    assert np.array_equal(s, keys)
    assert parse_index("OPQ128_1024,,PQ128", 1024) == (True, 1024, 128)
    assert parse_index("OPQ128_512,,PQ128", 1024) == (True, 512, 128)
    assert parse_index("OPQ64_512,PQ64", 512) == (True, 512, 64)
    assert parse_index("PQ64", 1024) == (False, 1024, 64)
    for bad in ("OPQ64_1024,,PQ128", "IVF4096,Flat", "OPQ128_2048,,PQ128", "PQ64x4"):
        with pytest.raises(ValueError):
            parse_index(bad, 1024)
