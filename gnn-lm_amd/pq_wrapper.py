"""``TorchPQCodec`` -- mirror of ``knn/pq_wrapper.py:91-203`` without faiss.

Buffers keep the reference's names (``centroids_torch [M,ksub,dsub]``, ``norm2_centroids_torch
[M,ksub]``, ``sdc_table_torch [M,ksub,ksub]``, optional ``A [d_out,d_in]`` / ``b [d_out]``), so the
``decoder.tgt_quantizer.*`` entries that ``fairseq_cli/convert_ckpt.py:40-45`` injects into a
checkpoint load with ``load_state_dict``.  The reference builds the tables from a faiss index
(``pq_wrapper.py:21-49``); faiss is not a dependency here, so the constructor takes the raw arrays
(``from_arrays`` / ``from_file`` on an ``.npz`` with keys centroids, A, b); a faiss index object is
still accepted when faiss is importable.

``decode`` is the hot-path half (transformer.py:1043-1045): table lookup by the HIP gather kernel,
then ``(x - b) @ A`` on the f32 MFMA GEMM (as ``x @ A - b @ A``).  ``encode`` (an offline tool in the
reference: quantize_features.py) runs on the GPU as GEMM + HIP argmin kernel for device
tensors; ``compute_sim`` is the plain torch expression of pq_wrapper.py:104-129.
"""
import numpy as np
import torch

from . import _lib


class TorchPQCodec(torch.nn.Module):
    def __init__(self, index=None, metric="ip", centroids=None, A=None, b=None):
        super().__init__()
        assert metric in ("l2", "ip")
        self.metric = metric
        if index is not None:
            centroids, A, b = self._from_faiss(index)
        if centroids is None:
            raise ValueError("TorchPQCodec needs a faiss index or the raw centroids")
        cen = torch.as_tensor(np.asarray(centroids), dtype=torch.float32)
        assert cen.dim() == 3 and cen.shape[1] == 256, "8-bit PQ codes only (pq_wrapper.py:33)"
        self.pre_torch = A is not None
        if self.pre_torch:
            self.register_buffer("A", torch.as_tensor(np.asarray(A), dtype=torch.float32))
            self.register_buffer("b", torch.as_tensor(np.asarray(b if b is not None else np.zeros(0)),
                                                      dtype=torch.float32))
        self.register_buffer("centroids_torch", cen)
        self.register_buffer("norm2_centroids_torch", (cen ** 2).sum(2))                         # :37
        if metric == "l2":                                                                         # :40-43
            sdc = -torch.sqrt(((cen[:, :, None, :] - cen[:, None, :, :]) ** 2).sum(3))
        else:                                                                                      # :46-48
            sdc = torch.matmul(cen, cen.transpose(1, 2))
        self.register_buffer("sdc_table_torch", sdc)
        self._prep = None

    @staticmethod
    def _from_faiss(index):
        import faiss                                                                               # optional
        A = b = None
        if isinstance(index, faiss.IndexPreTransform):
            vt = faiss.downcast_VectorTransform(index.chain.at(0))
            b = faiss.vector_to_array(vt.b)
            A = faiss.vector_to_array(vt.A).reshape(vt.d_out, vt.d_in)
            index = faiss.downcast_index(index.index)
        pq = index.pq
        return faiss.vector_to_array(pq.centroids).reshape(pq.M, pq.ksub, pq.dsub), A, b

    @classmethod
    def from_arrays(cls, centroids, A=None, b=None, metric="ip"):
        return cls(None, metric, centroids, A, b)

    @classmethod
    def from_file(cls, path, metric="ip"):
        """``.npz`` (keys centroids, A, b) or the reference's ``quantizer`` file itself: faiss's serialisation of
        ``IndexPreTransform(OPQMatrix -> IndexPQ)`` (quantize_features.py:108-109), read without faiss (faiss_io.py)."""
        if str(path).endswith(".npz"):
            z = np.load(path)
            return cls(None, metric, z["centroids"], z["A"] if "A" in z.files else None, z["b"] if "b" in z.files else None)
        from .faiss_io import read_pq_quantizer
        q = read_pq_quantizer(path)
        return cls(None, metric, q["centroids"], q["A"], q["b"])

    def save(self, path):
        arrs = {"centroids": self.centroids_torch.cpu().numpy()}
        if self.pre_torch:
            arrs.update(A=self.A.cpu().numpy(), b=self.b.cpu().numpy())
        np.savez(path, **arrs)

    # ------------------------------------------------------------------------------------------
    def decode(self, codes):
        """codes uint8 [n, M] (device) -> float32 [n, d_in]   (pq_wrapper.py:169-203)."""
        from . import ops
        n, MM = codes.shape
        M, ksub, dsub = self.centroids_torch.shape
        assert MM == M, f"input codes have {MM} subspace, but quantizer have {M} subspace"
        if not codes.is_cuda:
            raise _lib.GnnlmError("TorchPQCodec.decode runs on the GPU (HIP gather + MFMA GEMM); no CPU fallback")
        codes = codes.to(torch.uint8).contiguous()
        cen = self.centroids_torch
        if cen.device != codes.device:
            raise _lib.GnnlmError("move the codec to the codes' device first (.to(device))")
        if dsub % 4:
            raise NotImplementedError("dsub must be a multiple of 4 for the HIP decode kernel")
        x = ops.pq_lookup_direct(codes, cen)
        if not self.pre_torch:
            return x
        if self._prep is None or self._prep[0].device != codes.device:
            At = self.A.t().contiguous()                                            # [d_in, d_out]
            nba = -(self.b.double() @ self.A.double()).float() if self.b.numel() > 0 else None
            self._prep = (At, nba)
        At, nba = self._prep
        return ops.gemm_nt(x, At, bias=nba)

    def encode(self, x):
        """x [n, d_in] (device) -> codes uint8 [n, M]   (pq_wrapper.py:131-167): OPQ rotation on the f32 MFMA
        GEMM + the HIP argmin kernel (gnnlm_pq_encode).  Offline tool in the reference, a "next" row here."""
        if not x.is_cuda:
            raise _lib.GnnlmError("TorchPQCodec.encode runs on the GPU (MFMA GEMM + HIP argmin); no CPU fallback")
        from . import ops
        x = x.to(torch.float32).contiguous()
        if self.pre_torch:
            x = ops.gemm_nt(x, self.A.contiguous(), bias=self.b if self.b.numel() > 0 else None)
        M, ksub, dsub = self.centroids_torch.shape
        codes = torch.empty(x.shape[0], M, dtype=torch.uint8, device=x.device)
        _lib.call("gnnlm_pq_encode", _lib.ptr(x), x.stride(0), _lib.ptr(self.centroids_torch.contiguous()),
                  _lib.ptr(self.norm2_centroids_torch.contiguous()), M, dsub, x.shape[0], _lib.ptr(codes), _lib.stream())
        return codes

    def compute_sim(self, src, tgt):
        """sim[n, m] = sum_M sdc[M, src[n,M], tgt[m,M]]   (pq_wrapper.py:104-129).  Offline tool."""
        sdc = self.sdc_table_torch
        M = sdc.shape[0]
        ar = torch.arange(M, device=src.device)
        return sdc[ar[None, None, :], src.long()[:, None, :], tgt.long()[None, :, :]].sum(-1)
