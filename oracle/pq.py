"""Oracle: product-quantizer codec (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Restates ``knn/pq_wrapper.py`` of the reference in numpy:
  * tables      -- NumpyPQCodec.__init__  (pq_wrapper.py:14-49)
  * pq_encode   -- TorchPQCodec.encode    (pq_wrapper.py:131-167)
  * pq_decode   -- TorchPQCodec.decode    (pq_wrapper.py:169-203)
  * compute_sim -- TorchPQCodec.compute_sim (pq_wrapper.py:104-129)

Conventions (same as the reference): ``centroids [M, ksub, dsub]`` float32,
``A [d_out, d_in]`` / ``b [d_out]`` the OPQ linear pre-transform
(``encode`` applies ``x @ A.T + b``; ``decode`` applies ``(x - b) @ A``),
codes ``[n, M]`` uint8.
"""
import numpy as np


def pq_tables(centroids, metric="ip"):
    """norm2_centroids [M,ksub] and sdc_table [M,ksub,ksub] (pq_wrapper.py:37-49)."""
    cen = np.asarray(centroids, dtype=np.float32)
    norm2 = (cen ** 2).sum(axis=2)
    if metric == "l2":
        c1 = cen[:, :, None, :]
        c2 = cen[:, None, :, :]
        sdc = -np.sqrt(((c1 - c2) ** 2).sum(3))
    else:
        sdc = np.matmul(cen, cen.transpose(0, 2, 1))
    return norm2, sdc


def pq_encode(x, centroids, A=None, b=None):
    """codes[n,m] = argmin_c ||c||^2 - 2 x_m . c   (pq_wrapper.py:131-167)."""
    x = np.asarray(x, dtype=np.float32)
    if A is not None:
        x = x @ A.T
        if b is not None and b.size > 0:
            x = x + b
    n, d = x.shape
    M, ksub, dsub = centroids.shape
    assert d == M * dsub
    norm2 = (centroids ** 2).sum(axis=2)                       # [M, ksub]
    xs = x.reshape(n, M, dsub)
    dot = np.einsum("nmd,mkd->nmk", xs, centroids)             # [n, M, ksub]
    dis = norm2[None] - 2 * dot
    return dis.argmin(axis=2).astype(np.uint8)


def pq_lookup(codes, centroids):
    """x[n, m*dsub:(m+1)*dsub] = centroids[m, codes[n,m]]   (pq_wrapper.py:189-196).

    Pure table lookup: bit-exact by construction."""
    codes = np.asarray(codes)
    n, MM = codes.shape
    M, ksub, dsub = centroids.shape
    assert MM == M, f"input codes have {MM} subspace, but quantizer have {M} subspace"
    x = centroids[np.arange(M)[None, :], codes.astype(np.int64)]   # [n, M, dsub]
    return np.ascontiguousarray(x.reshape(n, M * dsub))


def pq_decode(codes, centroids, A=None, b=None):
    """TorchPQCodec.decode: lookup, then (x - b) @ A   (pq_wrapper.py:169-203)."""
    x = pq_lookup(codes, centroids).astype(np.float32)
    if A is not None:
        if b is not None and b.size > 0:
            x = x - b
        x = x @ A
    return x


def compute_sim(src_codes, tgt_codes, sdc_table):
    """sim[n,m] = sum_M sdc[M, src[n,M], tgt[m,M]]   (pq_wrapper.py:104-129)."""
    src = np.asarray(src_codes).astype(np.int64)
    tgt = np.asarray(tgt_codes).astype(np.int64)
    M = sdc_table.shape[0]
    sub = sdc_table[np.arange(M)[None, None, :], src[:, None, :], tgt[None, :, :]]
    return sub.sum(-1)
