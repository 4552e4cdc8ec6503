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

PARITY UNPINNED: no faiss-written file is available here to check the reader against; `write_pq_quantizer` emits the
same layout so that the pair round-trips (tests/test_mirrors_cpu.py)."""
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
        s = self.b[self.o:self.o + 4].decode("ascii", "replace")
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
