"""Reader (and, for tests / conversion, writer) of the one faiss serialisation the GNN-LM recipes need without faiss:
the ``quantizer`` file of ``knn/quantize_features.py:108-109`` -- ``faiss.write_index`` of an
``IndexPreTransform(OPQMatrix -> IndexPQ)`` (``"OPQ128_1024,,PQ128"``, gnnlm_scripts/wiki103/find_knn.sh:32-38) --
which ``NumpyPQCodec.__init__`` (knn/pq_wrapper.py:14-36) takes apart into ``A [d_out, d_in]``, ``b`` and
``centroids [M, 256, dsub]``.

faiss is not in this image and the reference pins no version, so the layout below is restated from faiss's published
``impl/index_write.cpp`` / ``index_read.cpp`` (little endian, ``WRITE1`` = raw scalar, ``WRITEVECTOR`` = uint64 count +
raw elements):

  index      := fourcc "IxPT" header  int32 n_transforms  transform*  index            (IndexPreTransform)
              | fourcc "IxPq" header  pq  vector<uint8> codes  int32 search_type  bool encode_signs  int32 polysemous_ht
  header     := int32 d  int64 ntotal  int64 dummy  int64 dummy  bool is_trained  int32 metric_type  [float metric_arg if metric_type > 1]
  transform  := fourcc ("LTra" | "rrot")  bool have_bias  vector<float> A  vector<float> b  int32 d_in  int32 d_out  bool is_trained
  pq         := uint64 d  uint64 M  uint64 nbits  vector<float> centroids              (size_t fields)

and the kNN index itself -- ``faiss.write_index`` of ``index_factory(d, "OPQ64_1024,IVF4096,PQ64", METRIC_INNER_PRODUCT)``
after ``add_with_ids`` (knn/index_builder.py:79-150), the file ``KNNModel`` is pointed at with ``--index-file``
(knn/knn_model.py:59-64):

  index      |= fourcc "IwPQ" ivf_header  bool by_residual  uint64 code_size  pq  invlists                (IndexIVFPQ)
              | fourcc ("IxFI" | "IxF2" | "IxFl") header  vector<float> xb                                 (IndexFlat, the coarse quantizer)
              | fourcc "IHNf" header  hnsw  index (an IndexFlat: the storage)                             (IndexHNSWFlat: the coarse quantizer of
                                                                                                             `IVF{n}_HNSW32`, what index_builder.py:60-64 picks
                                                                                                             by itself for >= 10^6 keys)
  hnsw       := vector<double> assign_probas  vector<int32> cum_nneighbor_per_level  vector<int32> levels  vector<uint64> offsets
                vector<int32> neighbors  int32 entry_point  int32 max_level  int32 efConstruction  int32 efSearch  int32 upper_beam
  ivf_header := header  uint64 nlist  uint64 nprobe  index (quantizer)  int8 direct_map_type  vector<int64> direct_map
                [vector<(int64, int64)> if direct_map_type == 2]
  index      |= fourcc ("IxMp" | "IxM2") header  index  vector<int64> id_map      (IndexIDMap[2]: `IDMap,,Flat`, what
                                                                                  index_builder.py:49-53 builds for < 30,000 keys)
  invlists   := fourcc "ilar"  uint64 nlist  uint64 code_size  fourcc ("full" | "sprs")  vector<uint64> sizes
                { uint8 codes[n * code_size]  int64 ids[n] } for every non-empty list          ("sprs": sizes = (list, n) pairs)

PARITY UNPINNED: no faiss-written file is available here to check the readers against; `write_pq_quantizer` /
`write_ivfpq_index` emit the same layouts so that the pairs round-trip (tests/test_mirrors_cpu.py)."""
import struct

import numpy as np


class _R:
    def __init__(self, data):
        self.b, self.o = data, 0

    def take(self, fmt):
        v = struct.unpack_from("<" + fmt, self.b, self.o)
        self.o += struct.calcsize("<" + fmt)
        return v[0] if len(v) == 1 else v

    def fourcc(self):
        s = bytes(self.b[self.o:self.o + 4]).decode("ascii", "replace")
        self.o += 4
        return s

    def vector(self, dtype):
        n = self.take("Q")
        a = np.frombuffer(self.b, dtype=dtype, count=n, offset=self.o).copy()
        self.o += n * np.dtype(dtype).itemsize
        return a


def _header(r):
    d, ntotal, _, _ = r.take("i"), r.take("q"), r.take("q"), r.take("q")
    trained, metric = r.take("?"), r.take("i")
    if metric > 1:
        r.take("f")
    return d, ntotal, trained, metric


def read_pq_quantizer(path):
    """-> dict(centroids f32 [M, ksub, dsub], A f32 [d_out, d_in] | None, b f32 [d_out] | None, metric "ip"|"l2")."""
    r = _R(open(path, "rb").read())
    A = b = None
    cc = r.fourcc()
    if cc == "IxPT":
        _header(r)
        nt = r.take("i")
        if nt != 1:
            raise ValueError(f"{path}: IndexPreTransform with {nt} transforms (the recipes use one OPQ matrix)")
        tcc = r.fourcc()
        if tcc not in ("LTra", "rrot"):
            raise ValueError(f"{path}: unsupported VectorTransform '{tcc}' (expected a LinearTransform / OPQMatrix)")
        have_bias = r.take("?")
        Av, bv = r.vector(np.float32), r.vector(np.float32)
        d_in, d_out, trained = r.take("i"), r.take("i"), r.take("?")
        if not trained or Av.size != d_in * d_out:
            raise ValueError(f"{path}: LinearTransform untrained or of inconsistent size")
        A = Av.reshape(d_out, d_in)                               # pq_wrapper.py:25
        b = bv if (have_bias and bv.size) else np.zeros(0, np.float32)
        cc = r.fourcc()
    if cc != "IxPq":
        raise ValueError(f"{path}: expected an IndexPQ ('IxPq'), found '{cc}'")
    d, _, trained, metric = _header(r)
    pd, M, nbits = r.take("Q"), r.take("Q"), r.take("Q")
    cen = r.vector(np.float32)
    if not trained or nbits != 8 or pd != d or d % M or cen.size != 256 * d:
        raise ValueError(f"{path}: need a trained 8-bit PQ (pq_wrapper.py:33), got d={pd} M={M} nbits={nbits}")
    return {"centroids": cen.reshape(M, 256, d // M), "A": A, "b": b, "metric": "ip" if metric == 0 else "l2"}


def write_pq_quantizer(path, centroids, A=None, b=None, metric="ip"):
    """The inverse, same layout (an empty IndexPQ: ntotal = 0, no codes)."""
    cen = np.ascontiguousarray(centroids, dtype=np.float32)
    M, ksub, dsub = cen.shape
    assert ksub == 256
    d = M * dsub
    mt = 0 if metric == "ip" else 1
    hdr = lambda dim: struct.pack("<iqqq?i", dim, 0, 1 << 20, 1 << 20, True, mt)
    vec = lambda a, dt: struct.pack("<Q", a.size) + np.ascontiguousarray(a, dtype=dt).tobytes()
    out = b""
    if A is not None:
        A = np.ascontiguousarray(A, dtype=np.float32)
        d_out, d_in = A.shape
        assert d_out == d
        bb = np.zeros(0, np.float32) if b is None else np.asarray(b, np.float32)
        out += b"IxPT" + hdr(d_in) + struct.pack("<i", 1)
        out += b"LTra" + struct.pack("<?", bb.size > 0) + vec(A, np.float32) + vec(bb, np.float32) + struct.pack("<ii?", d_in, d_out, True)
    out += b"IxPq" + hdr(d) + struct.pack("<QQQ", d, M, 8) + vec(cen, np.float32)
    out += vec(np.zeros(0, np.uint8), np.uint8) + struct.pack("<i?i", 0, False, 0)
    open(path, "wb").write(out)


def _linear_transform(r, path):
    tcc = r.fourcc()
    if tcc not in ("LTra", "rrot"):
        raise ValueError(f"{path}: unsupported VectorTransform '{tcc}' (expected a LinearTransform / OPQMatrix)")
    have_bias = r.take("?")
    Av, bv = r.vector(np.float32), r.vector(np.float32)
    d_in, d_out, trained = r.take("i"), r.take("i"), r.take("?")
    if not trained or Av.size != d_in * d_out:
        raise ValueError(f"{path}: LinearTransform untrained or of inconsistent size")
    return Av.reshape(d_out, d_in), (bv if (have_bias and bv.size) else None)


def sniff(path):
    """The fourcc a faiss index file starts with ('' if the file is too short)."""
    with open(path, "rb") as f:
        return f.read(4).decode("ascii", "replace")


def read_ivfpq_index(path):
    """``[IndexPreTransform(OPQ) ->] IndexIVFPQ`` -> dict(R [d_out, d_in] | None, coarse [nlist, d], pq [M, 256, dsub],
    list_off i64 [nlist + 1], list_ids i64 [N], list_codes u8 [N, M], nprobe, metric "ip" | "l2", by_residual).
    The file is memory-mapped; the lists are copied out one by one (codes and ids alternate in the file)."""
    r = _R(np.memmap(path, dtype=np.uint8, mode="r"))
    R = None
    cc = r.fourcc()
    if cc == "IxPT":
        _header(r)
        nt = r.take("i")
        if nt != 1:
            raise ValueError(f"{path}: IndexPreTransform with {nt} transforms (the recipes use one OPQ matrix)")
        R, b = _linear_transform(r, path)
        if b is not None and np.any(b):
            raise ValueError(f"{path}: LinearTransform with a bias (an OPQ matrix has none)")
        cc = r.fourcc()
    if cc != "IwPQ":
        raise ValueError(f"{path}: expected an IndexIVFPQ ('IwPQ'), found '{cc}'")
    d, ntotal, trained, metric = _header(r)
    nlist, nprobe = r.take("Q"), r.take("Q")
    qcc = r.fourcc()
    coarse_kind = "flat"
    if qcc == "IHNf":
        # IVF{n}_HNSW32: the centroids are the flat storage behind the HNSW graph.  The graph is skipped -- the device search
        # ranks ALL centroids exactly (a GEMM), where faiss walks the graph with efSearch candidates: the probed lists can differ
        # from faiss's approximate choice (they are the exact top-nprobe, never worse)
        _header(r)
        r.vector(np.float64); r.vector(np.int32); r.vector(np.int32); r.vector(np.uint64); r.vector(np.int32)
        r.take("i"); r.take("i"); r.take("i"); r.take("i"); r.take("i")
        qcc = r.fourcc()
        coarse_kind = "hnsw"
    if qcc not in ("IxFI", "IxF2", "IxFl"):
        raise ValueError(f"{path}: coarse quantizer '{qcc}' (expected an IndexFlat, or an IndexHNSWFlat over one)")
    qd, qn, _, qmetric = _header(r)
    xb = r.vector(np.float32)
    if qd != d or qn != nlist or xb.size != nlist * d:
        raise ValueError(f"{path}: coarse quantizer of inconsistent size")
    dm_type = r.take("b")
    r.vector(np.int64)
    if dm_type == 2:
        n = r.take("Q")
        r.o += 16 * n
    by_residual, code_size = r.take("?"), r.take("Q")
    pd, M, nbits = r.take("Q"), r.take("Q"), r.take("Q")
    cen = r.vector(np.float32)
    if not trained or nbits != 8 or pd != d or d % M or cen.size != 256 * d or code_size != M:
        raise ValueError(f"{path}: need a trained 8-bit IVF-PQ, got d={pd} M={M} nbits={nbits} code_size={code_size}")
    icc = r.fourcc()
    if icc != "ilar":
        raise ValueError(f"{path}: inverted lists '{icc}' (expected in-file array lists 'ilar')")
    il_n, il_cs = r.take("Q"), r.take("Q")
    kind = r.fourcc()
    sv = r.vector(np.uint64)
    sizes = np.zeros(nlist, np.int64)
    if kind == "full":
        sizes[:] = sv.astype(np.int64)
    elif kind == "sprs":
        sizes[sv[0::2].astype(np.int64)] = sv[1::2].astype(np.int64)
    else:
        raise ValueError(f"{path}: inverted list sizes '{kind}'")
    if il_n != nlist or il_cs != M or int(sizes.sum()) != ntotal:
        raise ValueError(f"{path}: inverted lists of inconsistent size")
    off = np.zeros(nlist + 1, np.int64)
    off[1:] = np.cumsum(sizes)
    codes = np.empty((ntotal, M), np.uint8)
    ids = np.empty(ntotal, np.int64)
    for l in range(nlist):
        n = int(sizes[l])
        if n:
            codes[off[l]:off[l + 1]] = np.frombuffer(r.b, dtype=np.uint8, count=n * M, offset=r.o).reshape(n, M)
            r.o += n * M
            ids[off[l]:off[l + 1]] = np.frombuffer(r.b, dtype=np.int64, count=n, offset=r.o)
            r.o += 8 * n
    return {"R": R, "coarse": xb.reshape(nlist, d), "pq": cen.reshape(M, 256, d // M), "list_off": off, "list_ids": ids,
            "list_codes": codes, "nprobe": int(nprobe), "metric": "ip" if metric == 0 else "l2",
            "coarse_metric": "ip" if qmetric == 0 else "l2", "by_residual": bool(by_residual), "coarse_kind": coarse_kind}


def write_ivfpq_index(path, R, coarse, pq, list_off, list_ids, list_codes, nprobe=1, metric="ip", coarse_kind="flat"):
    """The inverse, same layout (what ``faiss.write_index`` emits for ``OPQ..,IVF..,PQ..`` with array inverted lists).
    ``coarse_kind="hnsw"``: the coarse quantizer wrapped as an IndexHNSWFlat with an EMPTY graph (the layout of
    ``IVF{n}_HNSW32``; enough for the reader, which skips the graph -- not a file faiss could search)."""
    coarse = np.ascontiguousarray(coarse, dtype=np.float32)
    cen = np.ascontiguousarray(pq, dtype=np.float32)
    nlist, d = coarse.shape
    M = cen.shape[0]
    off = np.asarray(list_off, np.int64)
    N = int(off[-1])
    mt = 0 if metric == "ip" else 1
    hdr = lambda dim, n: struct.pack("<iqqq?i", dim, n, 1 << 20, 1 << 20, True, mt)
    vec = lambda a, dt: struct.pack("<Q", np.asarray(a).size) + np.ascontiguousarray(a, dtype=dt).tobytes()
    with open(path, "wb") as f:
        if R is not None:
            R = np.ascontiguousarray(R, dtype=np.float32)
            d_out, d_in = R.shape
            assert d_out == d
            f.write(b"IxPT" + hdr(d_in, N) + struct.pack("<i", 1))
            f.write(b"LTra" + struct.pack("<?", False) + vec(R, np.float32) + vec(np.zeros(0, np.float32), np.float32)
                    + struct.pack("<ii?", d_in, d_out, True))
        f.write(b"IwPQ" + hdr(d, N) + struct.pack("<QQ", nlist, nprobe))
        if coarse_kind == "hnsw":
            f.write(b"IHNf" + hdr(d, nlist) + vec(np.zeros(0), np.float64) + vec(np.zeros(0, np.int32), np.int32) * 2
                    + vec(np.zeros(0, np.uint64), np.uint64) + vec(np.zeros(0, np.int32), np.int32) + struct.pack("<iiiii", -1, -1, 40, 16, 1))
        f.write((b"IxFI" if mt == 0 else b"IxF2") + hdr(d, nlist) + vec(coarse, np.float32))
        f.write(struct.pack("<b", 0) + vec(np.zeros(0, np.int64), np.int64))
        f.write(struct.pack("<?Q", True, M) + struct.pack("<QQQ", d, M, 8) + vec(cen, np.float32))
        sizes = (off[1:] - off[:-1]).astype(np.uint64)
        f.write(b"ilar" + struct.pack("<QQ", nlist, M))
        nz = np.nonzero(sizes)[0]
        if nz.size > nlist // 2:
            f.write(b"full" + vec(sizes, np.uint64))
        else:
            f.write(b"sprs" + vec(np.stack([nz.astype(np.uint64), sizes[nz]], 1).reshape(-1), np.uint64))
        codes = np.asarray(list_codes, np.uint8)
        ids = np.asarray(list_ids, np.int64)
        for l in range(nlist):
            if sizes[l]:
                f.write(np.ascontiguousarray(codes[off[l]:off[l + 1]]).tobytes())
                f.write(np.ascontiguousarray(ids[off[l]:off[l + 1]]).tobytes())


def read_flat_index(path):
    """A faiss ``Flat`` index, bare or inside an ``IDMap`` (``index_factory(d, "IDMap,,Flat")``: the auto type of
    knn/index_builder.py:49-53 for datastores under 30,000 keys) -> dict(xb f32 [n, d], ids i64 [n] | None, metric "ip" | "l2")."""
    r = _R(np.memmap(path, dtype=np.uint8, mode="r"))
    cc = r.fourcc()
    mapped = cc in ("IxMp", "IxM2")
    if mapped:
        _header(r)
        cc = r.fourcc()
    if cc not in ("IxFI", "IxF2", "IxFl"):
        raise ValueError(f"{path}: not a Flat index ('{cc}')")
    d, ntotal, _, metric = _header(r)
    xb = r.vector(np.float32)
    if xb.size != d * ntotal:
        raise ValueError(f"{path}: Flat index holds {xb.size} floats, expected {ntotal} x {d}")
    ids = None
    if mapped:
        ids = r.vector(np.int64)
        if ids.size != ntotal:
            raise ValueError(f"{path}: id map of {ids.size} entries over {ntotal} vectors")
    return {"xb": xb.reshape(ntotal, d), "ids": ids, "metric": "ip" if metric == 0 else "l2"}


def write_flat_index(path, xb, ids=None, metric="ip"):
    """The inverse (``faiss.write_index`` of ``Flat`` / ``IDMap,,Flat``)."""
    xb = np.ascontiguousarray(xb, dtype=np.float32)
    n, d = xb.shape
    mt = 0 if metric == "ip" else 1
    hdr = struct.pack("<iqqq?i", d, n, 1 << 20, 1 << 20, True, mt)
    with open(path, "wb") as f:
        if ids is not None:
            f.write(b"IxMp" + hdr)
        f.write((b"IxFI" if mt == 0 else b"IxF2") + hdr + struct.pack("<Q", xb.size) + xb.tobytes())
        if ids is not None:
            ids = np.ascontiguousarray(ids, dtype=np.int64)
            f.write(struct.pack("<Q", ids.size) + ids.tobytes())
