"""``DataStore`` -- mirror of ``knn/data_store.py:21-102``: the (keys, vals) tables of one split.

Same constructor / ``from_pretrained`` signature, attributes (``keys, vals, dstore_size, hidden_size,
vocab_size, dstore_fp16, val_size, info``) and raw-memmap file format as the reference.  New here:
``to_device`` lifts the tables from host memmaps into HBM (the whole WikiText-103 label table is
413 MB, the fp16 key table 211 GB -- both fit one MI355X's 288 GB), which is what the hot path reads.
"""
import json
import logging
import os
import time

import numpy as np

LOGGING = logging.getLogger(__name__)


def vals_dtype(dstore_fp16, vocab_size):
    """int16 iff the store is fp16 and the vocabulary fits (data_store.py:50, eval_lm.py:202)."""
    return np.int16 if dstore_fp16 and vocab_size is not None and vocab_size < 2 ** 15 else np.int32


class DataStore:
    def __init__(self, dstore_size, hidden_size, dstore_dir, vocab_size=None, mode="r", dstore_fp16=False,
                 no_load_keys=False, use_memory=False, val_size=2):
        self.dstore_size, self.hidden_size, self.dstore_dir = dstore_size, hidden_size, dstore_dir
        self.vocab_size, self.no_load_keys = vocab_size, no_load_keys
        self.dstore_fp16, self.val_size = dstore_fp16, val_size
        os.makedirs(dstore_dir, exist_ok=True)
        if not no_load_keys:
            self.keys = np.memmap(os.path.join(dstore_dir, "keys.npy"), mode=mode,
                                  dtype=np.float16 if dstore_fp16 else np.float32, shape=(dstore_size, hidden_size))
        self.vals = np.memmap(os.path.join(dstore_dir, "vals.npy"), mode=mode,
                              dtype=vals_dtype(dstore_fp16, vocab_size) if vocab_size is not None else np.int32,
                              shape=(dstore_size, val_size))
        if val_size == 1:
            self.vals = self.vals.reshape(-1)
        if use_memory and mode == "r":
            t0 = time.time()
            if not no_load_keys:
                self.keys = np.array(self.keys)
            self.vals = np.array(self.vals)
            LOGGING.debug("Loading to memory took %.1f s", time.time() - t0)
        self._device_vals = None
        self._device_keys = None

    @property
    def info(self):
        return {"dstore_size": self.dstore_size, "hidden_size": self.hidden_size, "vocab_size": self.vocab_size,
                "dstore_fp16": self.dstore_fp16, "val_size": self.val_size}

    def save_info(self):
        with open(os.path.join(self.dstore_dir, "info.json"), "w") as f:
            json.dump(self.info, f, sort_keys=True, indent=4, ensure_ascii=False)

    @staticmethod
    def exists(dstore_dir):
        return all(os.path.exists(os.path.join(dstore_dir, n)) for n in ("keys.npy", "vals.npy"))

    @classmethod
    def from_pretrained(cls, dstore_dir, no_load_keys=False, use_memory=False, mode="r"):
        with open(os.path.join(dstore_dir, "info.json")) as f:
            info = json.load(f)
        return cls(dstore_size=info["dstore_size"], hidden_size=info["hidden_size"], dstore_dir=dstore_dir,
                   dstore_fp16=info.get("dstore_fp16", False), vocab_size=info.get("vocab_size", None),
                   no_load_keys=no_load_keys, mode=mode, use_memory=use_memory, val_size=info.get("val_size", 1))

    # ---- HBM residency ------------------------------------------------------------------------
    def vals_to_device(self, device, row0=0, n_rows=None):
        """Label table (or the shard [row0, row0+n_rows)) as an int16/int32 device tensor."""
        import torch
        assert self.val_size == 1, "make sure self.data_store.val_size == 1 (which is labels)"
        n_rows = self.dstore_size - row0 if n_rows is None else n_rows
        if self._device_vals is None or self._device_vals[0] != (str(device), row0, n_rows):
            t = torch.from_numpy(np.array(self.vals[row0:row0 + n_rows])).to(device)
            self._device_vals = ((str(device), row0, n_rows), t)
        return self._device_vals[1]

    def keys_to_device(self, device, chunk_rows=1 << 20):
        """Key table in HBM (chunked upload; 211 GB for WikiText-103 train -- check capacity first)."""
        import torch
        if self._device_keys is None or self._device_keys.device != torch.device(device):
            out = torch.empty(self.dstore_size, self.hidden_size, device=device,
                              dtype=torch.float16 if self.dstore_fp16 else torch.float32)
            for s in range(0, self.dstore_size, chunk_rows):
                e = min(self.dstore_size, s + chunk_rows)
                out[s:e] = torch.from_numpy(np.ascontiguousarray(self.keys[s:e]))
            self._device_keys = out
        return self._device_keys
