"""The owning HBM store of the C ABI (`gnnlm_store_*`, include/gnnlm.h) driven from ctypes the way a non-torch caller
would: the tables live in memory the LIBRARY allocated (hipMalloc inside gnnlm_store_create), filled by ranged
uploads from host buffers; `gnnlm_pq_gather_decode` and `gnnlm_hgt_forward` then read them through the raw pointers
`gnnlm_store_codes / _vals` return.  Results must equal the torch-owned-store path bit for bit."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_store_lifecycle_and_hgt_forward(dev):
    from gnnlm_amd import _lib, ops
    from gnnlm_amd.hgt import HGT, CodeStore, NeighborGraph
    L = _lib.lib()
    L.gnnlm_store_create.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                     ctypes.c_void_p]
    L.gnnlm_store_upload_codes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
    L.gnnlm_store_upload_vals.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
    L.gnnlm_store_codes.argtypes = [ctypes.c_void_p]
    L.gnnlm_store_vals.argtypes = [ctypes.c_void_p]
    L.gnnlm_store_destroy.argtypes = [ctypes.c_void_p]
    rs = np.random.RandomState(0)
    N, M, dsub, d, H, T, kg = 3000, 16, 4, 64, 4, 12, 6
    codes = rs.randint(0, 256, size=(N, M)).astype(np.uint8)
    vals = rs.randint(0, 200, size=N).astype(np.int16)
    cen = (rs.randn(M, 256, dsub) * 0.5).astype(np.float32)
    h = ctypes.c_void_p()
    _lib.check(L.gnnlm_store_create(N, 0, N, M, 2, 0, ctypes.byref(h)), "store_create")
    # ranged uploads, out of order, from host memory
    for lo, hi in [(1000, 3000), (0, 1000)]:
        _lib.check(L.gnnlm_store_upload_codes(h, codes[lo:hi].ctypes.data, lo, hi - lo, None), "upload_codes")
        _lib.check(L.gnnlm_store_upload_vals(h, vals[lo:hi].ctypes.data, lo, hi - lo, None), "upload_vals")
    assert L.gnnlm_store_upload_codes(h, codes.ctypes.data, N - 5, 10, None) != 0            # range past the shard: refused
    assert b"bad range" in L.gnnlm_last_error()
    p_codes, p_vals = L.gnnlm_store_codes(h), L.gnnlm_store_vals(h)
    assert p_codes and p_vals

    ids = rs.randint(0, N, size=(2 * T, kg)).astype(np.int64)
    ids[3] = -1
    ids_dev = torch.from_numpy(ids).to(dev)
    cen_dev = torch.from_numpy(cen).to(dev)
    # gather + decode + labels straight from the library-owned tables
    n_g = 5
    S = ids.size * n_g
    g = _lib.gnnlm_gather_t()
    x = torch.empty(S, M * dsub, device=dev)
    lab = torch.empty(S, device=dev, dtype=torch.int32)
    val = torch.empty(S, device=dev, dtype=torch.uint8)
    g.codes, g.vals, g.vals_itemsize = p_codes, p_vals, 2
    g.n_store, g.row0, g.n_local, g.M, g.dsub = N, 0, N, M, dsub
    g.centroids, g.ids, g.n_groups, g.left, g.right = cen_dev.data_ptr(), ids_dev.data_ptr(), ids.size, 2, 2
    g.out_x, g.ld_x, g.out_labels, g.out_valid = x.data_ptr(), M * dsub, lab.data_ptr(), val.data_ptr()
    _lib.call_desc("gnnlm_pq_gather_decode", g)
    ref = ops.pq_gather_decode(torch.from_numpy(codes).to(dev), cen_dev, ids_dev.reshape(-1), 2, 2,
                               vals=torch.from_numpy(vals).to(dev), want_labels=True)
    assert torch.equal(x, ref["x"]) and torch.equal(lab, ref["labels"]) and torch.equal(val, ref["valid"])

    # HGT forward with the descriptor pointed at the library-owned store
    torch.manual_seed(1)
    model = HGT(in_dim=d, hidden_dim=d, out_dim=d, n_layers=2, n_heads=H)
    tstore = CodeStore(codes=torch.from_numpy(codes).to(dev), centroids=cen_dev, n_store=N, vals=torch.from_numpy(vals).to(dev))
    tgt = torch.randn(2 * T, d, device=dev)
    G = NeighborGraph(ids=ids_dev, n_blocks=2, T=T, left=2, right=2, store=tstore)
    want = model(G, features={"tgt": tgt})["tgt"].clone()
    prep = model.prepare(tstore, dev)
    m = prep["model"]
    m.codes, m.vals, m.vals_itemsize = p_codes, p_vals, 2                       # swap in the raw pointers
    io = _lib.gnnlm_hgt_io_t()
    io.n_blocks, io.T, io.kg = 2, T, kg
    out = torch.empty_like(tgt)
    io.tgt_feats, io.ids, io.out_tgt = tgt.data_ptr(), ids_dev.data_ptr(), out.data_ptr()
    need = L.gnnlm_hgt_workspace_bytes(ctypes.byref(m), ctypes.byref(io))
    ws = torch.empty(need, device=dev, dtype=torch.uint8)
    _lib.check(L.gnnlm_hgt_forward(ctypes.byref(m), ctypes.byref(io), _lib.ptr(ws), need, _lib.stream()), "hgt_forward")
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    assert L.gnnlm_hgt_forward(ctypes.byref(m), ctypes.byref(io), _lib.ptr(ws), need - 1024, _lib.stream()) != 0   # short workspace
    _lib.check(L.gnnlm_store_destroy(h), "store_destroy")
    _lib.check(L.gnnlm_store_destroy(None), "store_destroy(NULL)")


def test_store_empty_shard_and_bad_arguments(dev):
    from gnnlm_amd import _lib
    L = _lib.lib()
    L.gnnlm_store_create.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                     ctypes.c_void_p]
    L.gnnlm_store_destroy.argtypes = [ctypes.c_void_p]
    h = ctypes.c_void_p()
    assert L.gnnlm_store_create(10, 10, 0, 8, 4, 0, ctypes.byref(h)) == 0          # more ranks than rows: an empty shard is legal
    assert L.gnnlm_store_destroy(h) == 0
    assert L.gnnlm_store_create(10, 5, 6, 8, 4, 0, ctypes.byref(h)) != 0           # shard past the end of the store
    assert L.gnnlm_store_create(10, 0, 10, 8, 3, 0, ctypes.byref(h)) != 0          # vals must be int16 / int32
