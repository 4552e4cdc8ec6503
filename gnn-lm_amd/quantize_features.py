#!/usr/bin/env python3
"""``quantize_features`` -- producer of the two biggest inputs of the hot path, mirror of ``knn/quantize_features.py:29-157``:

    python -m gnnlm_amd.quantize_features --data-dir D --subset train --index OPQ128_1024,,PQ128 --code-size 128 \\
        --chunk-size 10000000 [--compute-error] [--norm] [--pretrained_quantizer]

  1. gathers the reference's training sample (``--chunk-size`` rows taken as the FIRST rows of each of 100 equal parts of the
     key table, quantize_features.py:79-90; ``--norm`` L2-normalises them, :92-94),
  2. trains the quantizer the index string names -- ``OPQ<M>_<d_out>`` = an OPQ rotation ``A [d_out, d_in]`` (``ivfpq.train_opq``:
     alternating product-quantizer rounds and orthogonal Procrustes, what faiss's ``OPQMatrix`` does for the reference at :96,107),
     ``PQ<M>`` = M x 256 centroids by k-means on the rotated sample -- on the GPU,
  3. writes ``<data-dir>/quantizer[-norm]`` in faiss's own ``IndexPreTransform(OPQMatrix -> IndexPQ)`` serialisation
     (``faiss_io.write_pq_quantizer``; the reference: ``faiss.write_index`` at :108-110), so ``--quantizer_path`` /
     ``TorchPQCodec.from_file`` / the reference's ``faiss.read_index`` all read it,
  4. encodes EVERY key with ``TorchPQCodec.encode`` (OPQ rotation on the f32 MFMA GEMM + the HIP argmin kernel ``gnnlm_pq_encode``;
     the reference: :116-148 in batches of 8192) and
  5. writes ``<subset>_dstore/quantized-keys.npy`` as a real ``.npy`` (:150-152; ``language_modeling.py:276`` loads it with
     ``np.load``), and with ``--compute-error`` logs the reference's reconstruction error (mean over its 8192-row batches of
     ``|x - decode(codes)|^2 / |x|^2``, :137-148).

The reference delegates training to faiss (absent from this image: PARITY UNPINNED for the trained arrays themselves -- any
orthonormal ``A`` and any centroids give a valid codec; what is pinned is the FILE FORMATS, the sample, the encode arithmetic
against ``oracle/pq.py`` and the error definition).  An offline tool, GPU only: no CPU fallback."""
import argparse
import logging
import os
import re

import numpy as np
import torch

from . import ops
from .data_store import DataStore
from .faiss_io import write_pq_quantizer
from .ivfpq import _kmeans, train_opq
from .path_utils import dstore_path, quantized_feature_path, quantizer_path
from .pq_wrapper import TorchPQCodec

LOGGING = logging.getLogger("gnnlm_amd.quantize-features")


def get_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--data-dir", type=str, required=True, help="path to binary dataset directory")
    p.add_argument("--prefix", type=str, default="de-en", help="prefix of binary file (unused, as in the reference)")
    p.add_argument("--index", type=str, default="OPQ64_512,PQ64", help="quantizer index")
    p.add_argument("--subset", type=str, default="train", help="train/valid/test")
    p.add_argument("--code-size", type=int, default=64, help="bytes of quantized feature")
    p.add_argument("--chunk-size", type=int, default=10000000, help="maximum number of features to train")
    p.add_argument("--compute-error", action="store_true", default=False, help="compute reconstruction error")
    p.add_argument("--use-gpu", action="store_true", default=False, help="accepted for compatibility: this tool always runs on the GPU")
    p.add_argument("--norm", action="store_true", default=False, help="normalize feature vector to unit vector before quantize")
    p.add_argument("--pretrained_quantizer", action="store_true", default=False, help="use pretrained quantizer to encode features")
    # knobs the reference leaves to faiss's defaults
    p.add_argument("--opq-iters", type=int, default=10, help="OPQ rounds (PQ refit + Procrustes); 0 = random orthonormal rotation")
    p.add_argument("--pq-iters", type=int, default=25, help="k-means iterations of the final product quantizer (faiss: niter = 25)")
    p.add_argument("--opq-train", type=int, default=262144, help="rows of the sample the rotation is trained on")
    p.add_argument("--encode-rows", type=int, default=1 << 19, help="rows per upload + encode call (a multiple of 8192)")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--cuda", type=int, default=0)
    return p


def parse_index(s, hidden_size):
    """'OPQ128_1024,,PQ128' / 'OPQ64_512,PQ64' / 'PQ64' -> (opq: bool, d_out, M).  Empty fields between commas are legal faiss."""
    opq = re.search(r"OPQ(\d+)(?:_(\d+))?", s)
    pq = re.search(r"(?:^|,)PQ(\d+)(?:x(\d+))?", s)
    if pq is None or (pq.group(2) not in (None, "8")):
        raise ValueError(f"--index {s!r}: need a PQ<M> block with 8-bit codes (pq_wrapper.py:33)")
    M = int(pq.group(1))
    d_out = hidden_size
    if opq is not None:
        if int(opq.group(1)) != M:
            raise ValueError(f"--index {s!r}: OPQ{opq.group(1)} trains for {opq.group(1)} sub-spaces, the PQ block has {M}")
        d_out = int(opq.group(2)) if opq.group(2) else hidden_size
    if d_out % M or d_out > hidden_size:
        raise ValueError(f"--index {s!r}: d_out = {d_out} must divide into {M} sub-spaces and not exceed hidden_size = {hidden_size}")
    return opq is not None, d_out, M


def training_sample(keys, total_tokens, chunk_size):
    """quantize_features.py:79-90: the first ``part_size`` rows of each of 100 equal parts (the last part takes the remainder)."""
    n_train = min(chunk_size, total_tokens)
    num_parts = 100
    out = np.zeros([n_train, keys.shape[1]], dtype=np.float32)
    offset = 0
    for p_idx in range(num_parts):
        global_offset = total_tokens // num_parts * p_idx
        part_size = n_train // num_parts if p_idx < num_parts - 1 else n_train - n_train // num_parts * (num_parts - 1)
        rows = keys[global_offset: global_offset + part_size]                 # (the last part's remainder may reach past the table:
        out[offset: offset + len(rows)] = rows                                #  the reference's assignment raises there; here the
        offset += len(rows)                                                   #  sample is what exists)
    return out[:offset]


def _initial_rotation(x, d_out, gen):
    """[d_out, d_in] with orthonormal rows: a random rotation (d_out == d_in) or a random rotation of the d_out leading
    principal directions (d_out < d_in: what an OPQ<M>_<d_out> block starts from)."""
    d_in = x.shape[1]
    cpu_gen = torch.Generator().manual_seed(int(gen.initial_seed()))
    Q = torch.linalg.qr(torch.randn(d_out, d_out, generator=cpu_gen, dtype=torch.float64))[0]
    if d_out == d_in:
        return Q.to(torch.float32).to(x.device)
    xc = x.double()
    _, V = torch.linalg.eigh(xc.t() @ xc)                                     # ascending eigenvalues
    P = V[:, -d_out:].t().cpu()                                               # [d_out, d_in] leading directions
    return (Q @ P).to(torch.float32).to(x.device)


def train_quantizer(xt, opq, d_out, M, opq_iters, pq_iters, opq_train, seed):
    """xt [n, d_in] f32 on the device -> (centroids [M, 256, dsub], A [d_out, d_in] | None, b | None) as numpy."""
    dev = xt.device
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    A = None
    if opq:
        sub = xt if xt.shape[0] <= opq_train else xt[torch.randperm(xt.shape[0], generator=gen, device=dev)[:opq_train]]
        A = _initial_rotation(sub, d_out, gen)
        if opq_iters > 0:
            A = train_opq(sub, M, A, opq_iters, gen)
        xr = ops.gemm_nt(xt.contiguous(), A.contiguous())
    else:
        xr = xt
    dsub = d_out // M
    cen = torch.stack([_kmeans(xr[:, m * dsub:(m + 1) * dsub].contiguous(), 256, pq_iters, gen) for m in range(M)])
    return cen.cpu().numpy(), (A.cpu().numpy() if A is not None else None), (np.zeros(0, np.float32) if A is not None else None)


def main(args):
    if not torch.cuda.is_available():
        raise RuntimeError("gnnlm_amd.quantize_features needs a GPU (no CPU fallback)")
    dev = torch.device("cuda", max(args.cuda, 0))
    data_dir, subset, code_size = args.data_dir, args.subset, args.code_size
    ds = DataStore.from_pretrained(dstore_dir=dstore_path(data_dir=data_dir, subset=subset))
    info = ds.info
    hidden_size, total_tokens = info["hidden_size"], info["dstore_size"]

    if args.pretrained_quantizer:
        save_path = quantizer_path(data_dir)                                  # (as written at :57: without the -norm suffix)
        LOGGING.info(f"load pretrained quantizer at {save_path}")
        quantizer = TorchPQCodec.from_file(save_path).to(dev)
    else:
        opq, d_out, M = parse_index(args.index, hidden_size)
        if M != code_size:
            raise ValueError(f"--code-size {code_size} but --index {args.index!r} produces {M}-byte codes")
        LOGGING.info(f"Train quantized codes on first {args.chunk_size} features from")
        train_features = training_sample(ds.keys, total_tokens, args.chunk_size)
        if args.norm:
            train_features /= np.sqrt(np.sum(train_features ** 2, axis=-1, keepdims=True))
        LOGGING.info("Training Product Quantizer")
        cen, A, b = train_quantizer(torch.from_numpy(train_features).to(dev), opq, d_out, M, args.opq_iters, args.pq_iters,
                                    args.opq_train, args.seed)
        del train_features
        save_path = quantizer_path(data_dir, norm=args.norm)
        write_pq_quantizer(save_path, cen, A, b, metric="l2")                 # index_factory's default metric
        LOGGING.info(f"Save quantizer to {save_path}")
        quantizer = TorchPQCodec.from_arrays(cen, A, b).to(dev)
    if quantizer.centroids_torch.shape[0] != code_size:
        raise ValueError(f"--code-size {code_size} but the quantizer has {quantizer.centroids_torch.shape[0]} sub-spaces")

    qt_path = quantized_feature_path(data_dir, subset)
    # a real .npy, written through a memmap of the final file (the reference fills an in-RAM array and np.save()s it: same bytes)
    quantized_codes = np.lib.format.open_memmap(qt_path, mode="w+", dtype=np.uint8, shape=(total_tokens, code_size))
    bsz = 8192                                                                # the reference's batch: the unit of its error average
    rows = max(bsz, args.encode_rows // bsz * bsz)
    total_error = 0.0
    for start in range(0, total_tokens, rows):
        end = min(total_tokens, start + rows)
        x = torch.from_numpy(np.ascontiguousarray(ds.keys[start:end])).to(dev).to(torch.float32)
        if args.norm:
            x = x / (x ** 2).sum(-1, keepdim=True).sqrt()
        codes = quantizer.encode(x)
        if args.compute_error:
            x2 = quantizer.decode(codes)
            num, den = ((x - x2) ** 2).sum(1).double(), (x ** 2).sum(1).double()
            for s in range(0, end - start, bsz):                              # :139-141: one ratio of sums per 8192-row batch
                e = min(end - start, s + bsz)
                total_error += (num[s:e].sum() / den[s:e].sum()).item() * (e - s)
        quantized_codes[start:end] = codes.cpu().numpy()
    quantized_codes.flush()
    del quantized_codes
    out = {"quantizer": save_path, "quantized_keys": qt_path, "rows": total_tokens, "code_size": code_size}
    if args.compute_error:
        out["avg_reconstruction_error"] = total_error / max(total_tokens, 1)
        LOGGING.info(f"Avg Reconstruction error: {out['avg_reconstruction_error']}")
    LOGGING.info(f"Save quantized feature to {qt_path}")
    return out


if __name__ == "__main__":
    logging.basicConfig(format="%(asctime)s | %(levelname)s | %(name)s | %(message)s", datefmt="%Y-%m-%d %H:%M:%S", level=logging.INFO)
    main(get_parser().parse_args())
