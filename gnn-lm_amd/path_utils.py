"""On-disk layout of a GNN-LM data directory (mirror of ``knn/path_utils.py:13-40``).

    {data_dir}/{split}_dstore/info.json            dstore_size, hidden_size, vocab_size, dstore_fp16, val_size
    {data_dir}/{split}_dstore/keys.npy             RAW fp16/fp32 [N_split, d]   (no .npy header, data_store.py:44-47)
    {data_dir}/{split}_dstore/vals.npy             RAW int16/int32 [N_split, val_size]
    {data_dir}/train_dstore/quantized-keys.npy     real .npy, uint8 [N, M]      (quantize_features.py:152)
    {data_dir}/{split}_dstore/neighbors.mmap.{k}   RAW int64 [N_split, k], -1 = none (find_knn.py:56,66)
    {data_dir}/quantizer[-norm][suffix]            faiss index in the reference; here also quantizer.npz
"""
import os

_DSTORE = "{}_dstore"


def dstore_path(data_dir, subset):
    return os.path.join(data_dir, _DSTORE.format(subset))


def _in_dstore(data_dir, mode, name):
    return os.path.join(dstore_path(data_dir, mode), name)


def feature_path(data_dir, mode):
    return _in_dstore(data_dir, mode, "keys.npy")


def value_path(data_dir, mode):
    return _in_dstore(data_dir, mode, "vals.npy")


def quantized_feature_path(data_dir, mode):
    return _in_dstore(data_dir, mode, "quantized-keys.npy")


def neighbor_path(data_dir, mode, k=32):
    return _in_dstore(data_dir, mode, "neighbors.mmap.%d" % k)


def quantizer_path(data_dir, suffix="", norm=False):
    return os.path.join(data_dir, "quantizer" + ("-norm" if norm else "") + suffix)


def dictionary_path(data_dir):
    return os.path.join(data_dir, "dict.txt")


def fairseq_dataset_path(data_dir, mode):
    return os.path.join(data_dir, mode)
