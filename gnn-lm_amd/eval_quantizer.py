#!/usr/bin/env python3
"""``eval_quantizer`` -- mirror of ``knn/eval_quantizer.py:32-75``: the reconstruction error of the stored codes,

    mean over keys of |key - decode(quantized-keys[row])|^2 / |key|^2        (running mean over 1024-row chunks, :55-70)

for ``<data-dir>/quantizer`` + ``train_dstore/quantized-keys.npy`` against ``train_dstore/keys.npy``.  The decode is the hot
path's own (``TorchPQCodec.decode``: HIP table look-up + f32 MFMA GEMM).  GPU only."""
import argparse
import logging

import numpy as np
import torch

from .data_store import DataStore
from .path_utils import dstore_path, quantized_feature_path, quantizer_path
from .pq_wrapper import TorchPQCodec

LOGGING = logging.getLogger("gnnlm_amd.eval-quantizer")


def get_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--data-dir", type=str, required=True, help="path to binary dataset directory")
    p.add_argument("--use-gpu", action="store_true", default=False, help="accepted for compatibility: always on the GPU")
    p.add_argument("--subset", type=str, default="train")
    p.add_argument("--rows", type=int, default=1 << 18, help="rows per upload (the error is a per-row mean: chunking does not change it)")
    p.add_argument("--cuda", type=int, default=0)
    return p


def main(args):
    if not torch.cuda.is_available():
        raise RuntimeError("gnnlm_amd.eval_quantizer needs a GPU (no CPU fallback)")
    dev = torch.device("cuda", max(args.cuda, 0))
    save_path = quantizer_path(args.data_dir)
    LOGGING.info(f"load pretrained quantizer at {save_path}")
    quantizer = TorchPQCodec.from_file(save_path).to(dev)
    qt_codes = np.load(quantized_feature_path(args.data_dir, args.subset), mmap_mode="r")
    ds = DataStore.from_pretrained(dstore_dir=dstore_path(data_dir=args.data_dir, subset=args.subset))
    total = ds.keys.shape[0]
    assert total == qt_codes.shape[0]
    total_error = 0.0
    for offset in range(0, total, args.rows):
        end = min(offset + args.rows, total)
        fp = torch.from_numpy(np.ascontiguousarray(ds.keys[offset:end])).to(dev).to(torch.float32)
        qt = torch.from_numpy(np.ascontiguousarray(qt_codes[offset:end])).to(dev)
        rec = quantizer.decode(qt)
        total_error += (((fp - rec) ** 2).sum(-1).double() / (fp ** 2).sum(-1).double()).sum().item()
    err = total_error / max(total, 1)
    print(f"L2 error: {err}")
    return err


if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO)
    main(get_parser().parse_args())
