"""Oracle: IVF-PQ search with asymmetric distance computation (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Restates faiss's published IVFADC as the reference uses it (``knn/knn_model.py:87-101`` -> ``index.search`` on
``OPQ64_1024,IVF4096,PQ64``, inner product, residual codes, nprobe lists): numpy float64 accumulation over the SAME
index arrays the product searches.  faiss itself is absent from the image (parity unpinned against it)."""
import numpy as np


def search(q, R, coarse, pq, list_off, list_ids, list_codes, k, nprobe, cosine_queries=False, metric="ip"):
    """metric "ip": scores (inner products), best = largest, returned descending.  metric "l2" (``IndexBuilder``'s default,
    knn/index_builder.py:26,118): squared distances ``|q' - c_list - decode(code)|^2`` of faiss IVFADC with residual codes, the
    probed lists are the nprobe NEAREST centroids, best = smallest, returned ascending (missing: +inf)."""
    q = np.asarray(q, dtype=np.float64)
    if cosine_queries:
        q = q / np.sqrt((q ** 2).sum(1, keepdims=True))
    qr = q @ np.asarray(R, np.float64).T                                      # OPQ rotation
    M, _, dsub = pq.shape
    if metric == "l2":
        cen, pq64 = np.asarray(coarse, np.float64), np.asarray(pq, np.float64)
        d2c = (qr ** 2).sum(1)[:, None] - 2 * qr @ cen.T + (cen ** 2).sum(1)[None, :]
        out_v = np.full((len(q), k), np.inf)
        out_i = np.full((len(q), k), -1, dtype=np.int64)
        for r in range(len(q)):
            probes = np.argsort(d2c[r], kind="stable")[:nprobe]
            vs, is_ = [], []
            for l in probes:
                lo, hi = int(list_off[l]), int(list_off[l + 1])
                if hi > lo:
                    c = list_codes[lo:hi].astype(np.int64)
                    dec = pq64[np.arange(M)[None, :], c].reshape(hi - lo, M * dsub)       # decoded residuals
                    vs.append((((qr[r] - cen[l])[None, :] - dec) ** 2).sum(1))
                    is_.append(list_ids[lo:hi])
            if vs:
                v, i = np.concatenate(vs), np.concatenate(is_)
                top = np.lexsort((i, v))[:k]
                out_v[r, :len(top)], out_i[r, :len(top)] = v[top], i[top]
        return out_v, out_i
    cs = qr @ np.asarray(coarse, np.float64).T                                # coarse inner products
    lut = np.einsum("nmd,mcd->nmc", qr.reshape(len(q), M, dsub), np.asarray(pq, np.float64))     # ADC tables
    out_v = np.full((len(q), k), -np.inf)
    out_i = np.full((len(q), k), -1, dtype=np.int64)
    for r in range(len(q)):
        probes = np.argsort(-cs[r], kind="stable")[:nprobe]
        vs, is_ = [], []
        for l in probes:
            lo, hi = int(list_off[l]), int(list_off[l + 1])
            if hi > lo:
                c = list_codes[lo:hi].astype(np.int64)
                vs.append(cs[r, l] + lut[r][np.arange(M)[None, :], c].sum(1))
                is_.append(list_ids[lo:hi])
        if vs:
            v, i = np.concatenate(vs), np.concatenate(is_)
            top = np.lexsort((i, -v))[:k]
            out_v[r, :len(top)], out_i[r, :len(top)] = v[top], i[top]
    return out_v, out_i
