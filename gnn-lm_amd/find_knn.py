#!/usr/bin/env python3
"""``find_knn`` -- producer of ``{split}_dstore/neighbors.mmap.{k}``, mirror of ``knn/find_knn.py:28-70``
(a "next" row: the eval hot path only *reads* this file).

Same flags and file format (raw int64 ``[N_split, k]``, ``-1`` = none).  The reference searches a faiss
``faiss_store.cosine`` index of the candidate subset with the query subset's keys, WITHOUT normalising
the queries (find_knn.py:63-65 -- unlike knn_model.py:181-184; ranking by inner product is unaffected by
the query scale, appendix D.5); the reference keeps only the ids and drops the distances (:65-66) --
``--save-distances`` also writes them (``distances.mmap.{k}``, raw float32 ``[N_split, k]``; nothing in the
reference reads that file).  Here the search is exact on the GPU (``ExactIndex``: keys resident in HBM in
their stored dtype, chunked f32-MFMA GEMM + running top-k, no [n, N] matrix); a faiss index is used
instead when faiss is importable and the index file exists.  ``--truncate-to`` reproduces
``knn/truncate_neighbor_file.py:54`` (column truncation).
"""
import argparse
import logging
import os

import numpy as np
import torch

from .data_store import DataStore
from .knn_model import ExactIndex
from .path_utils import dstore_path, neighbor_path

LOGGING = logging.getLogger("gnnlm_amd.find_knn")


def get_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--data-dir", type=str, required=True)
    p.add_argument("--subset", type=str, required=True, help="train/valid/test, find knn of which subset")
    p.add_argument("--candidate_subset", type=str, default="train", help="find knn from which subset")
    p.add_argument("--cuda", default=0, type=int)
    p.add_argument("--nprobe", type=int, default=32)
    p.add_argument("--efsearch", type=int, default=8)
    p.add_argument("--k", type=int, default=32)
    p.add_argument("--bsz", type=int, default=1024)
    p.add_argument("--truncate-to", type=int, nargs="*", default=[], help="also write neighbors.mmap.{k'} for k' < k")
    p.add_argument("--save-distances", action="store_true", help="also write distances.mmap.{k} (float32)")
    return p


def open_index(args, device):
    index_file = os.path.join(dstore_path(args.data_dir, args.candidate_subset), "faiss_store.cosine")
    if os.path.exists(index_file + ".gnnlm.npz"):                               # this package's IVF-PQ index: searched on the GPU
        from .ivfpq import IVFPQIndex
        return IVFPQIndex.load(index_file + ".gnnlm.npz", device=device, nprobe=args.nprobe)
    try:
        import faiss
        if os.path.exists(index_file):
            index = faiss.read_index(index_file, faiss.IO_FLAG_ONDISK_SAME_DIR)
            faiss.ParameterSpace().set_index_parameter(index, "nprobe", args.nprobe)
            return index
    except ImportError:
        pass
    cand = DataStore.from_pretrained(dstore_path(args.data_dir, args.candidate_subset))
    LOGGING.info("exact search over %d candidate keys on %s", cand.dstore_size, device)
    return ExactIndex(cand.keys, "ip", cosine=True, device=device)            # index_builder.py:90-95,118


def main(args):
    if not torch.cuda.is_available():
        raise RuntimeError("gnnlm_amd.find_knn needs a GPU (no CPU fallback)")
    device = torch.device("cuda", max(args.cuda, 0))
    ds = DataStore.from_pretrained(dstore_path(args.data_dir, args.subset))
    index = open_index(args, device)
    out_file = neighbor_path(args.data_dir, args.subset, args.k)
    out = np.memmap(out_file, mode="w+", shape=(ds.dstore_size, args.k), dtype=np.int64)
    dist = None
    if args.save_distances:
        dist = np.memmap(os.path.join(os.path.dirname(out_file), f"distances.mmap.{args.k}"), mode="w+",
                         shape=(ds.dstore_size, args.k), dtype=np.float32)
    for start in range(0, ds.dstore_size, args.bsz):
        end = min(start + args.bsz, ds.dstore_size)
        q = np.asarray(ds.keys[start:end]).astype(np.float32)                    # not normalised, as written
        d, knns = index.search(q, args.k)
        out[start:end] = knns
        if dist is not None:
            dist[start:end] = d
    out.flush()
    if dist is not None:
        dist.flush()
    for k2 in args.truncate_to:
        assert k2 < args.k
        t = np.memmap(neighbor_path(args.data_dir, args.subset, k2), mode="w+", shape=(ds.dstore_size, k2), dtype=np.int64)
        t[:] = out[:, :k2]
        t.flush()
    print(f"Save neighbor of shape {out.shape} to {out_file}")
    return out_file


if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO)
    main(get_parser().parse_args())
