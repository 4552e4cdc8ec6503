"""The int8-MFMA filtered IVF-PQ scan (csrc/ivfpq_mfma.hip): the thresholded round of the on-device
replacement of faiss ``index.search`` (knn/knn_model.py:100; index ``OPQ64_1024,IVF4096,PQ64``, nprobe 32:
gnnlm_scripts/wiki103/find_knn.sh:8-13).

Contract under test: the filter is a SUPERSET of {score > tau} with a bounded excess, the re-score reproduces the float32
scan's bits, hence the search result is IDENTICAL to the one-pass float32 scan (``scan="f32"``) and agrees with the
float64 oracle ``oracle/ivfpq.py`` (faiss itself is absent from the image: parity unpinned against it)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def clustered(rs, N, d, n_centres, noise, zipf=False):
    centres = rs.randn(n_centres, d).astype(np.float32)
    if zipf:                                                 # cluster sizes ~ 1 / rank: skewed IVF lists
        p = 1.0 / np.arange(1, n_centres + 1)
        which = rs.choice(n_centres, size=N, p=p / p.sum())
    else:
        which = rs.randint(0, n_centres, N)
    return centres, (centres[which] + noise * rs.randn(N, d).astype(np.float32)).astype(np.float16)


def test_pack_tiles_layout(dev):
    from gnnlm_amd import ops
    rs = np.random.RandomState(0)
    N = 1000 + 7                                             # not a multiple of 16
    codes = rs.randint(0, 256, size=(N, 64)).astype(np.uint8)
    img = ops.ivfpq_pack_tiles(torch.from_numpy(codes).to(dev)).cpu().numpy().reshape(-1, 4, 16, 16)
    assert img.shape[0] == -(-N // 16)
    for r in (0, 1, 15, 16, 17, 500, N - 1):
        t, i = divmod(r, 16)
        for g in range(4):
            want = np.array([codes[r, 16 * g + (i + p) % 16] for p in range(16)], dtype=np.uint8)
            assert np.array_equal(img[t, g, i], want), (r, g)
    assert not img[-1, :, N % 16:, :].any()                  # rows beyond N are zero


def test_quantize_lut_bound(dev):
    """u8 tables with the one-sided bound the filter relies on: lo_m + u delta <~ L < lo_m + (u + 1) delta for every entry,
    incl. a constant sub-table, a flat query, huge offsets with a tiny range and a wide dynamic range between sub-tables."""
    from gnnlm_amd import ops
    rs = np.random.RandomState(1)
    n = 7
    lut = (rs.randn(n, 64, 256) * 0.05).astype(np.float32)
    lut[1, 5] = 0.25                                         # constant sub-table
    lut[2] = -3.0                                            # flat query: range 0
    lut[3] = 1000.0 + 1e-3 * rs.rand(64, 256)                # big offset, tiny range
    lut[4, :32] *= 100.0                                     # sub-tables of very different ranges
    lut[5, 7, 3] = 9.0                                       # one outlier entry sets delta
    qlut, qmeta = ops.ivfpq_quantize_lut(torch.from_numpy(lut.reshape(n, -1)).to(dev))
    qlut, qmeta = qlut.cpu().numpy(), qmeta.cpu().numpy()
    u = np.empty((n, 64, 256), np.int64)
    for m in range(64):
        u[:, m, :] = qlut[:, m // 32, :, m % 32].astype(np.int64) ^ 0x80        # stored as the signed byte u - 128
    L = lut.astype(np.float64)
    lo = L.min(2, keepdims=True)
    delta = qmeta[:, 0].astype(np.float64)[:, None, None]
    flat = np.array([False, False, True, False, False, False, False])     # query 2: range 0 (delta = 1e-30 vanishes next to -3 in float64)
    assert (L < lo + (u + 1) * delta)[~flat].all()           # the bound (exact arithmetic on the float32 values)
    assert (L >= lo + u * delta * (1 - 1e-4) - 1e-30).all()  # and it is tight: u is the floor
    np.testing.assert_allclose(qmeta[:, 1], lo.sum((1, 2)), rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(qmeta[:, 2], np.abs(lut).max((1, 2)))
    rng = (L.max(2) - L.min(2)).max(1)
    ok = rng > 0
    np.testing.assert_allclose(qmeta[ok, 0], rng[ok] / 255, rtol=2e-5)
    assert (u[2] == 0).all() and u.max() >= 254


def _prepare(index, q, dev):
    """The pieces of IVFPQIndex._search_block a direct call of the scan needs."""
    from gnnlm_amd import _lib, ops
    qr = ops.gemm_nt(q, index.R)
    cs = ops.gemm_nt(qr, index.coarse)
    nprobe = min(index.nprobe, index.nlist)
    pv = torch.empty(q.shape[0], nprobe, device=dev)
    pi = torch.empty(q.shape[0], nprobe, device=dev, dtype=torch.int64)
    ops.topk_merge(cs, pv, pi, largest=True, init=True)
    lut = torch.einsum("nmd,mcd->nmc", qr.view(q.shape[0], index.M, index.dsub), index.pq).reshape(q.shape[0], -1).contiguous()
    return cs, pi, lut


def _run_filter(index, q, tau, p_lo, dev, cap):
    from gnnlm_amd import _lib, ops
    cs, pi, lut = _prepare(index, q, dev)
    nq = q.shape[0]
    qlut, qmeta = ops.ivfpq_quantize_lut(lut, 64)
    grp_list, grp_q, n_groups, G = index._groups(pi[:, p_lo:])
    surv = torch.zeros(nq, cap, 2, device=dev, dtype=torch.int32)
    sc = torch.zeros(nq, 16, device=dev, dtype=torch.int32)                  # one 64-byte line per counter
    d = _lib.gnnlm_ivfpq_scan8_t()
    d.tiles, d.list_off, d.M = index.tiles.data_ptr(), index.list_off.data_ptr(), 64
    d.qlut, d.qmeta, d.coarse, d.ld_coarse, d.tau = qlut.data_ptr(), qmeta.data_ptr(), cs.data_ptr(), cs.stride(0), tau.data_ptr()
    d.grp_list, d.grp_q, d.n_groups, d.max_groups = grp_list.data_ptr(), grp_q.data_ptr(), n_groups.data_ptr(), G
    d.surv, d.surv_cnt, d.cap = surv.data_ptr(), sc.data_ptr(), cap
    # ABI 9: the entry point itself refuses an index its survivor records cannot address (18 bits of list, 19 of row)
    for nlist_, max_list_ in ((0, 0), (index.nlist, 1 << 19), ((1 << 18) + 1, index.max_list)):
        d.nlist, d.max_list = nlist_, max_list_
        with pytest.raises(_lib.GnnlmError, match="survivor records"):
            _lib.call_desc("gnnlm_ivfpq_scan8", d)
    d.nlist, d.max_list = index.nlist, index.max_list
    _lib.call_desc("gnnlm_ivfpq_scan8", d)
    torch.cuda.synchronize()
    return cs, pi, lut, qmeta, surv.cpu().numpy(), sc[:, 0].cpu().numpy(), (grp_list.cpu().numpy(), grp_q.cpu().numpy(), int(n_groups.item()))


@pytest.fixture(scope="module")
def small_index(dev):
    from gnnlm_amd.ivfpq import IVFPQIndex
    rs = np.random.RandomState(33)
    N, d, nlist = 150_001, 256, 24
    centres, keys = clustered(rs, N, d, 60, 0.6)
    index = IVFPQIndex.build(keys, nlist, 64, device=dev, cosine=True, nprobe=9, iters=5, seed=3)
    assert index.tiles is not None and index.packed_codes is None
    q = (centres[rs.randint(0, 60, 45)] + 0.6 * rs.randn(45, d)).astype(np.float32)
    q /= np.sqrt((q ** 2).sum(1, keepdims=True))
    return index, q


def test_groups(dev, small_index):
    """(query, probe) pairs -> groups of <= 8 queries of one list, sorted by list, every pair exactly once."""
    index, q = small_index
    rs = np.random.RandomState(5)
    pl = torch.from_numpy(np.stack([rs.permutation(index.nlist)[:7] for _ in range(37)])).to(dev)
    pl[3, 2] = -1
    pl[11, :] = -1
    grp_list, grp_q, n_groups, G = index._groups(pl)
    grp_list, grp_q, ng = grp_list.cpu().numpy(), grp_q.cpu().numpy(), int(n_groups.item())
    assert ng <= G and (grp_list[ng:] == -1).all() and (grp_q[ng:] == -1).all()
    live = grp_list[:ng]
    assert (np.diff(live[live >= 0]) >= 0).all()                             # sorted by list; the -1 bucket (if any) comes last
    got = sorted((int(l), int(qq)) for l, row in zip(grp_list[:ng], grp_q[:ng]) for qq in row if qq >= 0 and l >= 0)
    want = sorted((int(l), r) for r, row in enumerate(pl.cpu().numpy()) for l in row if l >= 0)
    assert got == want
    # full groups except the last one of a list
    for l in np.unique(live[live >= 0]):
        rows = grp_q[:ng][live == l]
        assert ((rows >= 0).sum(1)[:-1] == 8).all()
    # the device builder (gnnlm_ivfpq_build_groups) against the torch reference of the table: same lists, same group counts, same
    # queries per list (which queries share a group is left open), the segment offsets of the threshold pass, a strided probe table
    from gnnlm_amd.ivfpq import build_groups
    big = torch.from_numpy(np.stack([rs.permutation(index.nlist)[:9] for _ in range(300)])).to(dev)
    big[5, 1] = -1
    for tab, seg in ((pl, None), (big[:, :4], 1024), (big, 16)):
        dev_t, ref_t = index._groups(tab, seg), build_groups(tab, index.nlist, seg)
        ngd, ngr = int(dev_t[2].item()), int(ref_t[2].item())
        assert ngd == ngr and dev_t[3] == ref_t[3]
        gl, gq = dev_t[0].cpu().numpy(), dev_t[1].cpu().numpy()
        assert np.array_equal(gl, ref_t[0].cpu().numpy()) and (gq[ngd:] == -1).all()
        rq = ref_t[1].cpu().numpy()
        P = tab.shape[1]
        tab_h = tab.cpu().numpy()
        for l in np.unique(gl[:ngd]):
            assert sorted(gq[:ngd][gl[:ngd] == l].reshape(-1)) == sorted(rq[:ngr][gl[:ngr] == l].reshape(-1))
            assert ((gq[:ngd][gl[:ngd] == l] >= 0).sum(1)[:-1] == 8).all()
        if seg is not None:
            go = dev_t[4].cpu().numpy()
            assert (go[ngd:] == -1).all() and ((go >= 0) == (gq >= 0)).all()
            for g_ in range(ngd):
                for s_ in range(8):
                    if gq[g_, s_] >= 0:
                        slot = int(np.nonzero(tab_h[gq[g_, s_]] == gl[g_])[0][0])
                        assert go[g_, s_] == (gq[g_, s_] * P + slot) * seg


def test_filter_is_superset_with_bounded_excess(dev, small_index):
    """Every key of the probed lists with exact score > tau survives the int8 filter; what else survives lies within the
    quantisation band (64 + 2) delta + eps below tau.  tau per query: a quantile of its exact scores, -inf (all keys: the
    survivor count overflows the capacity and says so), +inf (none)."""
    index, q = small_index
    qd = torch.from_numpy(q).to(dev)
    nq = q.shape[0]
    arr = {a: getattr(index, a).cpu().numpy() for a in ("R", "coarse", "pq", "list_off", "list_codes")}
    qr = q.astype(np.float64) @ arr["R"].astype(np.float64).T
    cs64 = qr @ arr["coarse"].astype(np.float64).T
    lut64 = np.einsum("nmd,mcd->nmc", qr.reshape(nq, 64, -1), arr["pq"].astype(np.float64))
    cs, pi, lut = _prepare(index, qd, dev)
    pi_h = pi.cpu().numpy()
    p_lo = 2
    exact = []                                                               # per query: rows and float64 scores of lists p_lo..
    for r in range(nq):
        rows, sc = [], []
        for l in pi_h[r, p_lo:]:
            lo, hi = int(arr["list_off"][l]), int(arr["list_off"][l + 1])
            c = arr["list_codes"][lo:hi].astype(np.int64)
            rows.append(np.arange(lo, hi))
            sc.append(cs64[r, l] + lut64[r][np.arange(64)[None, :], c].sum(1))
        exact.append((np.concatenate(rows), np.concatenate(sc)))
    tau = np.array([np.quantile(exact[r][1], 0.99) for r in range(nq)], dtype=np.float32)
    tau[0], tau[1] = -np.inf, np.inf
    cap = 4096
    cs_, pi_, lut_, qmeta, surv, sc, _ = _run_filter(index, qd, torch.from_numpy(tau).to(dev), p_lo, dev, cap)
    qmeta = qmeta.cpu().numpy()
    from gnnlm_amd import ops
    u_tab = ops.ivfpq_quantize_lut(lut_, 64)[0].cpu().numpy().astype(np.int64) ^ 0x80     # [nq, 2, 256, 32]: the byte tables
    assert sc[0] == len(exact[0][0]) and sc[0] > cap                         # tau = -inf: everything, counted beyond the capacity
    assert sc[1] == 0
    list_of_row = np.searchsorted(arr["list_off"], np.arange(index.ntotal), side="right") - 1
    for r in range(2, nq):
        rows, s64 = exact[r]
        assert sc[r] <= cap
        got = surv[r, :sc[r]]
        got_rows = got[:, 0].astype(np.int64) & 0xffffffff
        assert len(np.unique(got_rows)) == len(got_rows)                     # no key twice
        assert np.array_equal(list_of_row[got_rows], got[:, 1].astype(np.int64) & 0x3ffff)   # {row, list | sum_u << 18} records
        su = (got[:, 1].astype(np.int64) & 0xffffffff) >> 18                 # ... whose sum_u is the key's exact integer sum
        exact_su = sum(u_tab[r, m // 32, arr["list_codes"][got_rows, m].astype(np.int64), m % 32] for m in range(64))
        small = su < 16383                                                   # (16383: beyond what the filter's staging entry holds)
        assert np.array_equal(su[small], exact_su[small]) and small.mean() > 0.9
        must = rows[s64 > tau[r] - 1e-6]                                     # (1e-6: float32 vs float64 scores at the threshold)
        assert np.isin(must, got_rows).all(), r                              # superset
        band = 66.0 * qmeta[r, 0] + 1e-4
        may = rows[s64 > tau[r] - band]
        assert np.isin(got_rows, may).all(), r                               # bounded excess
        assert len(got_rows) < 4 * len(must) + 64, (r, len(got_rows), len(must))


def test_threshold_pass_is_a_lower_bound(dev, small_index):
    """The threshold pass (integer sums of the first D lists -> histogram -> tau): at least k keys of those lists score above
    tau (so the k-th best of the whole search does), and tau is within the quantisation band of their exact k-th best."""
    from gnnlm_amd import _lib, ops
    index, q = small_index
    qd = torch.from_numpy(q).to(dev)
    nq = q.shape[0]
    cs, pi, lut = _prepare(index, qd, dev)
    nprobe = pi.shape[1]
    pv = torch.gather(cs, 1, pi)
    qlut, qmeta = ops.ivfpq_quantize_lut(lut, 64)
    arr = {a: getattr(index, a).cpu().numpy() for a in ("R", "coarse", "pq", "list_off", "list_codes")}
    qr = q.astype(np.float64) @ arr["R"].astype(np.float64).T
    cs64 = qr @ arr["coarse"].astype(np.float64).T
    lut64 = np.einsum("nmd,mcd->nmc", qr.reshape(nq, 64, -1), arr["pq"].astype(np.float64))
    pi_h = pi.cpu().numpy()
    u = (qlut.cpu().numpy().astype(np.int64) ^ 0x80)                           # [nq, 2, 256, 32]: the byte tables
    for D, k in ((3, 1024), (2, 64), (1, 9000)):
        hist = torch.zeros(nq, D, 1024, device=dev, dtype=torch.int32)
        index._scan8(qlut, qmeta, cs, index._groups(pi[:, :D], seg=1024), hist=hist)
        tau = torch.empty(nq, device=dev)
        t = _lib.gnnlm_ivfpq_tau_t()
        t.hist, t.D = hist.data_ptr(), D
        t.probe_list, t.probe_bias, t.ld_probe = pi.data_ptr(), pv.data_ptr(), pi.stride(0)
        t.qmeta, t.n, t.k, t.tau = qmeta.data_ptr(), nq, k, tau.data_ptr()
        _lib.call_desc("gnnlm_ivfpq_tau", t)
        tau_h, hist_h, dl = tau.cpu().numpy(), hist.cpu().numpy(), qmeta[:, 0].cpu().numpy()
        for r in range(nq):
            sc = []
            for d_, l in enumerate(pi_h[r, :D]):
                lo, hi = int(arr["list_off"][l]), int(arr["list_off"][l + 1])
                c = arr["list_codes"][lo:hi].astype(np.int64)
                sc.append(cs64[r, l] + lut64[r][np.arange(64)[None, :], c].sum(1))
                if r % 7 == 0:                                               # the histogram: sum_m u of every key of the list, exactly
                    su = sum(u[r, m // 32, c[:, m], m % 32] for m in range(64))
                    assert np.array_equal(hist_h[r, d_], np.bincount(su >> 4, minlength=1024)), (D, r, d_)
            sc = np.concatenate(sc)
            if len(sc) < k:
                assert np.isneginf(tau_h[r])
                continue
            assert (sc > tau_h[r]).sum() >= k, (D, k, r)                      # a valid threshold
            kth = np.sort(sc)[-k]
            assert tau_h[r] > kth - 112 * dl[r] - 1e-4, (D, k, r, tau_h[r], kth, dl[r])   # and not a loose one (64 + bins of 16 twice + slack)


def test_rescore_long_survivor_lists(dev, small_index):
    """gnnlm_ivfpq_rescore on its own: 0, 1, 1023 .. 40,000 survivors per query (past the 16 unrolled steps of 1024 survivors: the
    rolled tail), every score against the float64 sum and the payload of every candidate."""
    from gnnlm_amd import _lib
    index, q = small_index
    qd = torch.from_numpy(q[:8]).to(dev)
    nq, cap = 8, 40960
    cs, pi, lut = _prepare(index, qd, dev)
    rs = np.random.RandomState(12)
    off = index.list_off.cpu().numpy()
    counts = [0, 1, 1023, 1025, 16384, 16385, 20000, 40000]
    surv = np.zeros((nq, cap, 2), np.int32)
    rows_of = []
    for r, n in enumerate(counts):
        rows = rs.choice(index.ntotal, size=n, replace=False) if n else np.zeros(0, np.int64)
        surv[r, :n, 0] = rows
        surv[r, :n, 1] = np.searchsorted(off, rows, side="right") - 1
        rows_of.append(rows)
    sc16 = np.zeros((nq, 16), np.int32)
    sc16[:, 0] = counts
    tau = torch.full((nq,), float("-inf"), device=dev)
    cv = torch.zeros(nq, cap, device=dev)
    ci = torch.full((nq, cap), -7, device=dev, dtype=torch.int64)
    cc = torch.zeros(nq, device=dev, dtype=torch.int32)
    surv_d, sc_d = torch.from_numpy(surv).to(dev), torch.from_numpy(sc16).to(dev)
    r_ = _lib.gnnlm_ivfpq_rescore_t()
    r_.codes, r_.payload, r_.M = index.list_codes.data_ptr(), index.payload.data_ptr(), 64
    r_.lut, r_.ld_lut, r_.coarse, r_.ld_coarse, r_.tau = lut.data_ptr(), lut.stride(0), cs.data_ptr(), cs.stride(0), tau.data_ptr()
    r_.surv, r_.surv_cnt, r_.cap, r_.n = surv_d.data_ptr(), sc_d.data_ptr(), cap, nq
    r_.cand_val, r_.cand_id, r_.cand_cnt, r_.cand_cap = cv.data_ptr(), ci.data_ptr(), cc.data_ptr(), cap
    _lib.call_desc("gnnlm_ivfpq_rescore", r_)
    assert cc.cpu().numpy().tolist() == counts                               # tau = -inf: every survivor is a candidate
    lut_h = lut.cpu().numpy().astype(np.float64).reshape(nq, 64, 256)
    cs_h, codes, payload = cs.cpu().numpy().astype(np.float64), index.list_codes.cpu().numpy().astype(np.int64), index.payload.cpu().numpy()
    cv_h, ci_h = cv.cpu().numpy(), ci.cpu().numpy()
    for r, n in enumerate(counts):
        rows = rows_of[r]
        ref = cs_h[r, surv[r, :n, 1]] + lut_h[r][np.arange(64)[None, :], codes[rows]].sum(1)
        by_id = dict(zip(payload[rows].tolist(), ref.tolist()))
        assert sorted(ci_h[r, :n].tolist()) == sorted(payload[rows].tolist()) and (ci_h[r, n:] == -7).all()
        got_ref = np.array([by_id[i] for i in ci_h[r, :n].tolist()])
        np.testing.assert_allclose(cv_h[r, :n], got_ref, rtol=2e-5, atol=2e-5)


def _assert_same(va, ia, vb, ib, what=None):
    """Scores bit-identical; ids identical except inside runs of exactly equal scores (keys with identical codes): the
    float32 dense round breaks such ties by list position, the candidate merge by id."""
    assert np.array_equal(va, vb), (float((va != vb).mean()), np.nonzero((va != vb).any(1))[0][:10], what)
    diff = ia != ib
    if diff.any():
        r, c = np.nonzero(diff)
        tied = (va[r, np.maximum(c - 1, 0)] == va[r, c]) | (va[r, np.minimum(c + 1, va.shape[1] - 1)] == va[r, c])
        assert tied.all(), (float(diff.mean()), r[:10], c[:10], what)
        assert diff.mean() < 0.01, (float(diff.mean()), what)


def _same_search(a, b, q, k):
    va, ia = a.search(q, k)
    vb, ib = b.search(q, k)
    _assert_same(va, ia, vb, ib, (a.cand_cap, b.cand_cap))
    return va, ia


def test_mfma_search_identical_to_f32_scan(dev, small_index):
    """The whole search through the int8 filter + re-score == the one-pass float32 scan, bit for bit (ids and scores), for
    k = 1024 / 64, one dense list and tight candidate capacities (overflowing queries are searched again on their own)."""
    from gnnlm_amd.ivfpq import IVFPQIndex
    from oracle import ivfpq as oivf
    index, q = small_index
    args = (index.R, index.coarse, index.pq, index.list_off, index.list_ids, index.list_codes)
    f32 = IVFPQIndex(*args, nprobe=9, scan="f32")
    assert f32.tiles is None and f32.packed_codes is not None
    for k in (1024, 64):
        v, i = _same_search(index, f32, q, k)
        arrs = [getattr(index, a).cpu().numpy() for a in ("R", "coarse", "pq", "list_off", "list_ids", "list_codes")]
        v_ref, i_ref = oivf.search(q, *arrs, k=k, nprobe=9)
        assert np.mean([len(set(a) & set(b)) / k for a, b in zip(i, i_ref)]) > 0.998
        np.testing.assert_allclose(v, v_ref, rtol=2e-5, atol=2e-5)
    one = IVFPQIndex(*args, nprobe=9, dense_probes=1)
    _same_search(one, IVFPQIndex(*args, nprobe=9, dense_probes=1, scan="f32"), q, 1024)
    tight = IVFPQIndex(*args, nprobe=9, cand_cap=64)
    rs = np.random.RandomState(2)
    qr = rs.randn(9, q.shape[1]).astype(np.float32)                          # unclustered queries: neighbours in every probed list
    qr /= np.sqrt((qr ** 2).sum(1, keepdims=True))
    _same_search(tight, f32, np.concatenate([q[:8], qr]), 1024)
    # a few overflowing queries among many: only those are searched again (the capacity stays)
    mixed = IVFPQIndex(*args, nprobe=9, cand_cap=2048)
    many = np.concatenate([q, q[::-1], qr[:2]])
    _same_search(mixed, f32, many, 1024)


def test_sampled_threshold_pass_is_verified(dev, small_index):
    """The threshold pass over a SAMPLE of its lists' tiles (ABI 8, `sums_stride`): the search result is the exact one whatever the sample
    says -- a query left with fewer than k candidates above its threshold is searched again with an exact threshold pass (few of them), or
    the index stops sampling (many of them)."""
    import torch
    from gnnlm_amd.ivfpq import IVFPQIndex
    index, q = small_index
    args = (index.R, index.coarse, index.pq, index.list_off, index.list_ids, index.list_codes)
    f32 = IVFPQIndex(*args, nprobe=9, scan="f32")
    qd = np.concatenate([q, q[::-1]])
    for k in (1024, 64):
        for S in (2, 4, 8):
            a = IVFPQIndex(*args, nprobe=9)
            a.threshold_sample, a.sample_min_keys_per_k = S, 0                  # sample whatever the lists hold
            _same_search(a, f32, qd, k)
            assert a.threshold_sample == S and "underflow" in a.stats
    # a margin far below zero: the sampled rank is 1, the threshold too high for most queries
    b = IVFPQIndex(*args, nprobe=9)
    b.threshold_sample, b.sample_min_keys_per_k, b.sample_sigmas, b.sample_fail_frac = 4, 0, -1e9, 1.0
    _same_search(b, f32, qd, 1024)
    assert b.threshold_sample == 4 and int(b.stats["requeried"]) > 0 and float(b.stats["underflow"]) > 0   # searched again one by one
    c = IVFPQIndex(*args, nprobe=9)
    c.threshold_sample, c.sample_min_keys_per_k, c.sample_sigmas = 4, 0, -1e9
    _same_search(c, f32, qd, 1024)
    assert c.threshold_sample == 1                                             # too many of them: no more sampling on this index


def test_search_repeats_bit_for_bit(dev, small_index):
    """The filter's records reach a query's list in any order (atomics between workgroups, ranks from LDS atomics inside one, waves that
    run ahead into the next group): nothing downstream may depend on it -- 20 searches of the same batch return the same bits
    (`tools/search_repeat.py` does the same at the 103 M-key shape)."""
    import torch
    index, q = small_index
    qd = torch.from_numpy(np.concatenate([q, q[::-1]])).to(dev)
    v0, i0 = index.search_device(qd, 1024)
    v0, i0 = v0.clone(), i0.clone()
    for _ in range(20):
        v, i = index.search_device(qd, 1024)
        assert torch.equal(v, v0) and torch.equal(i, i0)


def test_empty_list_among_the_threshold_lists(dev, small_index):
    """A query whose best probes include EMPTY lists: the threshold pass writes no histogram for an empty list, so the tau kernel
    must not read stale memory there (a too-high tau silently drops true neighbours).  Empty lists are added as scaled-up copies of
    the centroids (they outrank every real list); the search runs twice with dirtied allocator blocks in between."""
    from gnnlm_amd.ivfpq import IVFPQIndex
    from oracle import ivfpq as oivf
    index, q = small_index
    extra = 4
    N = index.ntotal
    coarse = torch.cat([index.coarse, 1.5 * index.coarse[:extra]]).contiguous()
    list_off = torch.cat([index.list_off, torch.full((extra,), N, dtype=torch.int64, device=dev)]).contiguous()
    emp = IVFPQIndex(index.R, coarse, index.pq, list_off, index.list_ids, index.list_codes, nprobe=9)
    assert emp.tiles is not None and emp.dense_probes == 6
    f32 = IVFPQIndex(index.R, coarse, index.pq, list_off, index.list_ids, index.list_codes, nprobe=9, scan="f32", dense_probes=6)
    qd = torch.from_numpy(q).to(dev)
    cs = (qd @ index.R.t()) @ coarse.t()
    top = cs.topk(6, dim=1).indices
    assert bool((top >= index.nlist).any(1).sum() > 5)                       # several queries do have an empty list among their 6 best
    arrs = [a.cpu().numpy() for a in (index.R, coarse, index.pq, list_off, index.list_ids, index.list_codes)]
    v_ref, i_ref = oivf.search(q, *arrs, k=1024, nprobe=9)
    for attempt in range(3):
        junk = [torch.full((q.shape[0], 6, 1024), 1 << 20, device=dev, dtype=torch.int32) for _ in range(3)]   # what a stale histogram looks like
        del junk
        v, i = emp.search_device(qd, 1024)
        v, i = v.cpu().numpy(), i.cpu().numpy()
        fin = np.isfinite(v_ref)
        assert np.array_equal(np.isfinite(v), fin), attempt
        np.testing.assert_allclose(v[fin], v_ref[fin], rtol=2e-5, atol=2e-5)
        assert np.mean([len(set(a) & set(b)) / 1024 for a, b in zip(i, i_ref)]) > 0.998
    vf, if_ = f32.search_device(qd, 1024)
    _assert_same(v, i, vf.cpu().numpy(), if_.cpu().numpy())


def test_search_result_does_not_depend_on_the_grouping(dev, small_index):
    """Which queries of a list share a workgroup, and in which order survivors and candidates are appended, is decided by
    atomics (gnnlm_ivfpq_build_groups, the survivor counters): the RESULT must not depend on it -- repeated searches, the torch
    reference of the task table and a shuffled query order all give the same scores and ids, bit for bit."""
    import os
    index, q = small_index
    qd = torch.from_numpy(np.concatenate([q, q[::-1], q[5:25]])).to(dev)
    v0, i0 = index.search_device(qd, 1024)
    for _ in range(3):
        v, i = index.search_device(qd, 1024)
        assert torch.equal(v, v0) and torch.equal(i, i0)
    os.environ["GNNLM_IVF_TORCH_GROUPS"] = "1"
    try:
        v, i = index.search_device(qd, 1024)
    finally:
        del os.environ["GNNLM_IVF_TORCH_GROUPS"]
    assert torch.equal(v, v0) and torch.equal(i, i0)
    perm = torch.randperm(qd.shape[0], generator=torch.Generator().manual_seed(3)).to(dev)
    v, i = index.search_device(qd[perm].contiguous(), 1024)
    assert torch.equal(v, v0[perm]) and torch.equal(i, i0[perm])


def test_labels_travel_with_the_search(dev, small_index):
    """attach_vals: the search returns vals[ids] with the neighbours (knn/knn_model.py:198 without the gather), ids and
    scores unchanged, -1 padding reads the last label like numpy's wrap-around."""
    from gnnlm_amd.ivfpq import IVFPQIndex
    index, q = small_index
    rs = np.random.RandomState(8)
    vals = rs.randint(0, 267_744, index.ntotal).astype(np.int32)
    args = (index.R, index.coarse, index.pq, index.list_off, index.list_ids, index.list_codes)
    lab = IVFPQIndex(*args, nprobe=9).attach_vals(torch.from_numpy(vals).to(dev))
    qd = torch.from_numpy(q).to(dev)
    for k in (1024, 16):
        v0, i0 = index.search_device(qd, k)
        v1, i1, kv = lab.search_device(qd, k, return_vals=True)
        assert torch.equal(v0, v1) and torch.equal(i0, i1)
        assert np.array_equal(kv.cpu().numpy(), vals[i1.cpu().numpy()])
    # fewer stored keys than k: -1 ids read vals[-1]
    tiny = IVFPQIndex.build(rs.randn(300, 256).astype(np.float32), 8, 64, device=dev, cosine=False, nprobe=3, iters=4, seed=0)
    tv = rs.randint(0, 1000, 300).astype(np.int16)
    tiny.attach_vals(torch.from_numpy(tv).to(dev))
    v, i, kv = tiny.search_device(torch.from_numpy(rs.randn(5, 256).astype(np.float32)).to(dev), 200, return_vals=True)
    i, kv = i.cpu().numpy(), kv.cpu().numpy()
    assert (i < 0).any() and (i >= 0).any()
    assert np.array_equal(kv, np.where(i >= 0, tv[np.maximum(i, 0)], tv[-1]))


@pytest.fixture(scope="module")
def reference_shape_index(dev):
    """The reference's index shape (find_knn.sh:8-13): d = 1024, OPQ64_1024, IVF4096, PQ64 (dsub 16), over 8.4 M clustered
    fp16 keys generated on the device with Zipf cluster sizes (skewed list lengths)."""
    from gnnlm_amd.ivfpq import IVFPQIndex
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    N, d, nc = 8_400_000, 1024, 3000
    centres = torch.randn(nc, d, generator=g, device=dev)
    p = 1.0 / torch.arange(1, nc + 1, device=dev, dtype=torch.float64) ** 0.7
    which = torch.multinomial(p / p.sum(), N, replacement=True, generator=g)
    keys = torch.empty(N, d, device=dev, dtype=torch.float16)
    for s in range(0, N, 1 << 19):
        w = which[s:s + (1 << 19)]
        keys[s:s + (1 << 19)] = (centres[w] + 0.8 * torch.randn(w.shape[0], d, generator=g, device=dev)).to(torch.float16)
    index = IVFPQIndex.build(keys, 4096, 64, device=dev, cosine=True, nprobe=32, iters=6, seed=5)
    qi = torch.randint(0, nc, (300,), generator=g, device=dev)
    q = centres[qi] + 0.8 * torch.randn(300, d, generator=g, device=dev)
    q = q / q.norm(dim=1, keepdim=True)
    return index, q, keys


def test_reference_shape_search_vs_oracle(dev, reference_shape_index, tmp_path):
    """`OPQ64_1024,IVF4096,PQ64`, nprobe 32, k = 1024 (knn_model.py:28,100; find_knn.sh:8-13) over 8.4 M keys with skewed
    lists: the device search == the float64 IVFADC oracle over the same index arrays (ids as sets >= 0.998, scores 2e-5),
    == the one-pass float32 scan bit for bit, and the same through KNNModel(--index-file) after a faiss-format round trip."""
    import json, os
    from gnnlm_amd import faiss_io
    from gnnlm_amd.ivfpq import IVFPQIndex
    from gnnlm_amd.knn_model import KNNModel
    from oracle import ivfpq as oivf, knn as oknn
    index, q, keys = reference_shape_index
    lens = (index.list_off[1:] - index.list_off[:-1]).cpu().numpy()
    assert index.d == 1024 and index.M == 64 and index.dsub == 16 and index.nlist == 4096 and index.tiles is not None
    print("list lengths: min %d median %d max %d" % (lens.min(), np.median(lens), lens.max()))
    assert lens.max() > 1.5 * np.median(lens) and lens.min() < 0.67 * np.median(lens)    # skewed lists
    k = 1024
    v, i = index.search_device(q, k)
    v, i = v.cpu().numpy(), i.cpu().numpy()
    arrs = [getattr(index, a).cpu().numpy() for a in ("R", "coarse", "pq", "list_off", "list_ids", "list_codes")]
    qh = q.cpu().numpy()
    v_ref, i_ref = oivf.search(qh, *arrs, k=k, nprobe=32)
    same = np.mean([len(set(a) & set(b)) / k for a, b in zip(i, i_ref)])
    assert same > 0.998, same
    np.testing.assert_allclose(v, v_ref, rtol=2e-5, atol=2e-5)
    assert (np.diff(v, axis=1) <= 0).all() and (i >= 0).all()
    f32 = IVFPQIndex(index.R, index.coarse, index.pq, index.list_off, index.list_ids, index.list_codes, nprobe=32, scan="f32")
    v2, i2 = f32.search_device(q, k)
    _assert_same(v, i, v2.cpu().numpy(), i2.cpu().numpy())
    del f32
    # the index as the reference's own file: faiss `IndexPreTransform(OPQ) -> IndexIVFPQ` written, read back by KNNModel
    V = 5000
    vals = np.random.RandomState(3).randint(0, V, index.ntotal).astype(np.int16)
    dd = tmp_path / "train_dstore"
    os.makedirs(dd)
    vals.tofile(dd / "vals.npy")
    json.dump({"dstore_size": index.ntotal, "hidden_size": 1024, "vocab_size": V, "dstore_fp16": True, "val_size": 1}, open(dd / "info.json", "w"))
    f = str(dd / "faiss_store.cosine")
    faiss_io.write_ivfpq_index(f, *arrs, nprobe=1)
    m = KNNModel(f, str(dd), k=k, probe=32, no_load_keys=True, metric_type="do_not_recomp_ip", device=dev)
    assert isinstance(m.index, IVFPQIndex) and m.index.nprobe == 32 and m.index.has_vals and m.index.tiles is not None
    raw = q * torch.linspace(0.5, 3.0, q.shape[0], device=dev)[:, None]      # KNNModel normalises the queries (knn_model.py:181-184)
    targets = torch.from_numpy(vals[i[:, 3]].astype(np.int64)).to(dev)        # a retrieved neighbour's label: recall > 0
    p, rec = m.get_knn_prob(raw, targets=targets, t=0.05, return_recall=True)
    p_ref, rec_ref = oknn.knn_target_prob(v_ref.astype(np.float32), i_ref, vals, targets.cpu().numpy(), 0.05)
    np.testing.assert_allclose(p.cpu().numpy(), p_ref.numpy(), rtol=2e-3, atol=1e-6)
    assert np.abs(rec.cpu().numpy() - rec_ref.numpy()).max() <= 2            # near-ties at the k-th place may swap a neighbour
    assert rec.min().item() >= 1


def test_refine_tightens_the_threshold_and_keeps_the_result(dev, small_index):
    """gnnlm_ivfpq_refine (ABI 7): the k-th largest LOWER bound among a query's survivors is a valid threshold (at least k keys of
    the probed lists score above it), never below the threshold pass's, the records dropped cannot beat it, and the search result is
    the one without the step -- bit for bit."""
    from gnnlm_amd.ivfpq import IVFPQIndex
    index, q = small_index
    args = (index.R, index.coarse, index.pq, index.list_off, index.list_ids, index.list_codes)
    a, b = IVFPQIndex(*args, nprobe=9), IVFPQIndex(*args, nprobe=9)
    b.refine_tau = False
    a.keep_candidates = b.keep_candidates = True
    qd = torch.from_numpy(np.concatenate([q, q[::-1]])).to(dev)
    for k in (1024, 64, 2000):
        va, ia = a.search_device(qd, k)
        tau_a, cnt_a = a.last_candidates[3].cpu().numpy(), a.last_candidates[2].cpu().numpy()
        vb, ib = b.search_device(qd, k)
        tau_b, cnt_b = b.last_candidates[3].cpu().numpy(), b.last_candidates[2].cpu().numpy()
        assert torch.equal(va, vb) and torch.equal(ia, ib), k
        assert (tau_a >= tau_b).all() and (cnt_a <= cnt_b).all()
        assert (cnt_a >= np.minimum(k, cnt_b)).all()                          # still at least k candidates wherever there were k
        if k == 1024:
            assert cnt_a.sum() < 0.9 * cnt_b.sum(), (k, cnt_a.sum(), cnt_b.sum())   # and fewer of them to score and select from (6 of this
                                                                                    # index's 9 probed lists are threshold lists: little is lost to begin with)
        sa, sb = float(a.stats["rescored"]), float(b.stats["survivors"])
        assert sa <= sb and float(a.stats["survivors"]) == sb
