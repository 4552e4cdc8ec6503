"""Parity at the FULL store sizes of BASELINE.json configs[4] and configs[3] (GPU only; one MI355X holds either store).

  configs[4] One Billion Word: 800,000,000 keys x 128-byte PQ codes = 102.4 GB in HBM (+ 3.2 GB of int32 labels), PQ
      128 x 256 x 4 behind a 512 x 1024 OPQ (`OPQ128_512,,PQ128`, gnnlm_scripts/one_billion/find_knn.sh:24-30), vocabulary
      793,471, 2 HGT layers: rows spread over the whole table -- byte offsets up to 1.0e11, rows > 7e8, the last row --
      through `gnnlm_pq_gather_decode`, `gnnlm_star_attn` (PQ source, dsub = 4) and `gnnlm_knn_interp` against the oracle
      on exactly those rows, then a 48-token block with 2 layers against the un-elided float64 oracle.
  configs[3] EnWik8: 90,000,000 keys x 64-byte codes (PQ 64 x 256 x 8, 512 x 512 OPQ, d = 512), int16 labels (205
      characters): the 4-GiB byte-offset boundary sits at row 2^26 < 90 M -- rows on both sides of it, the last row, int16
      labels at high rows, and a 256-token block against the float64 oracle.
The stores are generated on the device with the generator bench.py uses (synthetic.device_codes); the oracle only ever
sees the rows a test touches (oracle/hostrows.py).  Reference behaviour: token_block_dataset.py:338-412,
pq_wrapper.py:169-203, hgt.py:299-420, knn_model.py:192-217, data_store.py:44-52."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import graph as og
from oracle import hgt as ohgt
from oracle import knn as oknn
from oracle import pq as opq
from oracle.hostrows import HostRows

CFG = {
    "one_billion": dict(N=800_000_000, M=128, dsub=4, d=1024, H=8, vocab=793_471, vals=torch.int32, T=48, L=2, seed=41,
                        special=[2 ** 25, 2 ** 25 + 1, 2 ** 29 + 3, 700_000_001, 799_999_998]),
    "enwik8": dict(N=90_000_000, M=64, dsub=8, d=512, H=8, vocab=205, vals=torch.int16, T=256, L=1, seed=43,
                   special=[2 ** 26 - 1, 2 ** 26, 2 ** 26 + 1, 80_000_001, 89_999_998]),
}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module", params=list(CFG))
def store(request, dev):
    from gnnlm_amd.synthetic import device_codes, make_codec, zipf_dev
    c = CFG[request.param]
    free = torch.cuda.mem_get_info(dev)[0]
    need = c["N"] * (c["M"] + 8)
    if need > 0.9 * free:
        pytest.skip(f"{request.param}: the {need / 2 ** 30:.0f} GiB store does not fit the {free / 2 ** 30:.0f} GiB of free HBM")
    codes = device_codes(c["N"], c["M"], dev, c["seed"])
    g = torch.Generator(device=dev)
    g.manual_seed(c["seed"] + 1)
    vals = torch.empty(c["N"], dtype=c["vals"], device=dev)
    for s in range(0, c["N"], 1 << 26):                                       # in pieces: the generator works in int64
        e = min(c["N"], s + (1 << 26))
        vals[s:e] = zipf_dev(e - s, c["vocab"], g, dev).to(c["vals"])
    cen, A, b = make_codec(np.random.RandomState(c["seed"]), c["M"], c["dsub"], c["d"], opq=True)
    yield dict(c, name=request.param, codes=codes, vals_dev=vals, cen=cen, A=A, b=b)
    del codes, vals
    torch.cuda.empty_cache()


def spread_rows(rs, n, N, special):
    fixed = [0, 1, 2] + list(special) + [N - 3, N - 2, N - 1]
    hi = rs.randint(N // 2, N, size=n // 2)
    lo = rs.randint(0, N, size=n - n // 2 - len(fixed))
    return np.concatenate([fixed, hi, lo]).astype(np.int64)


def test_gather_decode_full_size(dev, store):
    from gnnlm_amd import ops
    N, M = store["N"], store["M"]
    rs = np.random.RandomState(1)
    ids = np.concatenate([spread_rows(rs, 8000, N, store["special"]), [-1, N, N + 7]])
    rows, valid = og.slot_layout(ids.reshape(-1, 1), N, 2, 2)
    rows, valid = rows.reshape(-1), valid.reshape(-1)
    host = HostRows(store["codes"], rows[valid])
    out = ops.pq_gather_decode(store["codes"], torch.from_numpy(store["cen"]).to(dev), torch.from_numpy(ids).to(dev), 2, 2,
                               vals=store["vals_dev"], want_codes=True, want_labels=True)
    assert np.array_equal(out["valid"].cpu().numpy().astype(bool), valid)
    assert (rows[valid].astype(np.float64) * M >= 2.0 ** 32).sum() > 10_000        # byte offsets past 4 GiB
    assert np.array_equal(out["codes"].cpu().numpy()[valid], host[rows[valid]])     # bytes: exact
    x = out["x"].cpu().numpy()
    assert np.array_equal(x[valid], opq.pq_lookup(host[rows[valid]], store["cen"]))  # table look-up: exact
    assert not x[~valid].any()
    lab = out["labels"].cpu().numpy()
    vh = store["vals_dev"][torch.from_numpy(rows[valid]).to(dev)].cpu().numpy()
    assert np.array_equal(lab[valid], vh) and (lab[~valid] == -1).all()


def test_star_attn_full_size(dev, store):
    from gnnlm_amd import ops
    from tests.test_fullsize_gpu import star_reference
    N, H, dpq = store["N"], store["H"], store["M"] * store["dsub"]
    rs = np.random.RandomState(2)
    T, kg = 64, 128
    ids = spread_rows(rs, T * kg, N, store["special"]).reshape(T, kg)
    ids[0, 5], ids[1, :], ids[2, 7], ids[2, 9] = -1, -1, N, N + 12345
    ok = (ids >= 0) & (ids < N)
    host = HostRows(store["codes"], ids[ok])
    U = (rs.randn(T, H, dpq) / np.sqrt(dpq)).astype(np.float32)
    Z, has = ops.star_attn(torch.from_numpy(U).to(dev), torch.from_numpy(ids).to(dev), codes=store["codes"],
                           centroids=torch.from_numpy(store["cen"]).to(dev))
    X = np.zeros((T, kg, dpq))
    X[ok] = opq.pq_lookup(host[ids[ok]], store["cen"])
    ref = star_reference(U, ids, X, ok)
    assert np.abs(Z.cpu().numpy() - ref).max() < 2e-5
    assert np.array_equal(has.cpu().numpy(), ok.any(1).astype(np.float32))


def test_knn_interp_full_size(dev, store):
    from gnnlm_amd import ops
    N, V = store["N"], store["vocab"]
    rs = np.random.RandomState(3)
    n, k = 128, 1024
    ids = spread_rows(rs, n * k, N, store["special"]).reshape(n, k)
    ids[:, :4] = rs.randint(int(0.875 * N), N, size=(n, 4))                   # labels near the end of the table in every row
    ids[::5, -3:] = -1                                                        # padding: wraps to the LAST row like numpy
    vals = store["vals_dev"].cpu().numpy()                                   # 3.2 GB for One Billion Word: the oracle indexes it like the reference does
    sims = np.sort(rs.uniform(0.2, 0.9, size=(n, k)).astype(np.float32), axis=1)[:, ::-1].copy()
    targets = np.where(rs.rand(n) < 0.5, vals[ids[:, 2]], rs.randint(0, V, size=n)).astype(np.int64)
    lm = np.log(rs.uniform(1e-4, 1, size=n)).astype(np.float32)
    for t, lmbda in [(0.01, 0.1), (1.0, 0.25)]:
        p_ref, rec_ref = oknn.knn_target_prob(sims, ids, vals, targets, t)
        ref = oknn.combine_knn_and_vocab_probs(p_ref, torch.from_numpy(lm), lmbda)
        out, pk, rec = ops.knn_interp(*(torch.from_numpy(a).to(dev) for a in (lm, sims, ids, targets)), t, lmbda, vals=store["vals_dev"])
        assert np.array_equal(rec.cpu().numpy(), rec_ref.numpy())
        np.testing.assert_allclose(pk.cpu().numpy(), p_ref.numpy(), rtol=5e-5, atol=1e-7)
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=2e-5, atol=5e-6)


def test_block_full_size(dev, store):
    """One block (48 tokens x 2 layers for One Billion Word, 256 tokens x 1 layer for EnWik8), k_g = 128, l = r = 2, neighbours
    over the whole store, against the un-elided float64 oracle."""
    from gnnlm_amd.hgt import CodeStore
    from tests.test_hgt_gpu import run_hip
    N, d, H, L, T = store["N"], store["d"], store["H"], store["L"], store["T"]
    rs = np.random.RandomState(4)
    kg = 128
    nb = rs.randint(0, N, size=(T, kg)).astype(np.int64)
    nb[rs.rand(T, kg) < 0.001] = -1
    nb[3] = -1
    nb[0, :4 + len(store["special"])] = [0, 1, N - 1, N - 2] + list(store["special"])     # clipped contexts, offsets past 4 GiB
    tgt = rs.randn(T, d).astype(np.float16).astype(np.float32)
    rows, valid = og.slot_layout(nb, N, 2, 2)
    host = HostRows(store["codes"], rows[valid])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cs = CodeStore(codes=store["codes"], centroids=t(store["cen"]), n_store=N, vals=store["vals_dev"], A=t(store["A"]), b=t(store["b"]))
    sd = {k: v.numpy() for k, v in ohgt.init_hgt_weights(L, d, H, seed=9).items()}
    out = run_hip(dev, sd, L, H, d, cs, nb, 1, T, 2, 2, tgt, return_ntgt=False)["tgt"]
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 128)))
    gr = og.build_graph(nb, np.zeros(T, np.int64), N, 2, 2)
    ntgt = opq.pq_lookup(host[gr["ntgt_offsets"]], store["cen"]).astype(np.float64)
    ntgt = (ntgt - store["b"].astype(np.float64)) @ store["A"].astype(np.float64)
    feats = {"tgt": torch.from_numpy(tgt.astype(np.float64)), "ntgt": torch.from_numpy(ntgt)}
    ref = ohgt.hgt_forward({k: torch.as_tensor(v).double() for k, v in sd.items()}, L, H, feats, gr)["tgt"].numpy()
    err = np.abs(out - ref).max()
    assert np.isfinite(out).all() and err < 2e-4, (store["name"], err)
