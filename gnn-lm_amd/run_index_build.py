#!/usr/bin/env python3
"""``run_index_build`` -- producer of the kNN index, mirror of ``knn/run_index_build.py`` + ``knn/index_builder.py`` for the
index family the recipes use (``--index-type OPQ64_1024,IVF4096,PQ64 --metric cosine``,
gnnlm_scripts/wiki103/find_knn.sh:8-13).  The reference hands training and adding to faiss; here both run on the GPU
(``IVFPQIndex.build``) and the result is written next to where the reference puts its file:
``<dstore-dir>/faiss_store.<metric><suffix>.gnnlm.npz`` (``KNNModel`` / ``find_knn`` pick it up from the reference's
``--index-file`` path).  An offline tool: a "next" row of SURVEY.md 8f, not the eval hot path."""
import argparse
import logging
import os
import re

import torch

from .data_store import DataStore
from .ivfpq import IVFPQIndex

LOGGING = logging.getLogger("gnnlm_amd.run_index_build")


def get_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--dstore-dir", type=str, required=True)
    p.add_argument("--index-type", type=str, default="OPQ64_1024,IVF4096,PQ64")
    p.add_argument("--metric", type=str, default="cosine", choices=["l2", "ip", "cosine"],
                   help="l2: squared distances (the reference's own default, knn/run_index_build.py:50); the recipes pass cosine")
    p.add_argument("--suffix", type=str, default="")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--max-train", type=int, default=1000000)
    p.add_argument("--chunk-size", type=int, default=1 << 18)
    p.add_argument("--nprobe", type=int, default=32)
    p.add_argument("--opq-iters", type=int, default=10, help="rounds of OPQ training when --index-type has an OPQ block (0: random rotation)")
    p.add_argument("--overwrite", action="store_true")
    p.add_argument("--cuda", type=int, default=0)
    return p


def parse_index_type(s):
    """'OPQ64_1024,IVF4096,PQ64' -> (nlist, M); the OPQ block only fixes the rotation's shape (square here)."""
    ivf, pq = re.search(r"IVF(\d+)", s), re.search(r"(?:^|,)PQ(\d+)", s)
    if not (ivf and pq):
        raise ValueError(f"--index-type {s!r}: only OPQ*,IVF<nlist>,PQ<M> indexes are built")
    return int(ivf.group(1)), int(pq.group(1))


def main(args):
    if not torch.cuda.is_available():
        raise RuntimeError("gnnlm_amd.run_index_build needs a GPU (no CPU fallback)")
    out = os.path.join(args.dstore_dir, f"faiss_store.{args.metric}{args.suffix}.gnnlm.npz")
    if os.path.exists(out) and not args.overwrite:
        LOGGING.info("%s exists, use --overwrite to rebuild", out)
        return out
    ds = DataStore.from_pretrained(args.dstore_dir)
    nlist, M = parse_index_type(args.index_type)
    nlist = max(1, min(nlist, ds.dstore_size // 30 or 1))                       # index_builder.py:56: at least ~30 keys per list
    index = IVFPQIndex.build(ds.keys, nlist, M, device=torch.device("cuda", max(args.cuda, 0)), cosine=(args.metric == "cosine"),
                             nprobe=args.nprobe, train_size=args.max_train, seed=args.seed, chunk=args.chunk_size,
                             opq_iters=args.opq_iters if "OPQ" in args.index_type else 0, metric="l2" if args.metric == "l2" else "ip")
    index.save(out)
    print(f"Save index of {index.ntotal} keys ({nlist} lists, PQ{M}) to {out}")
    return out


if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO)
    main(get_parser().parse_args())
