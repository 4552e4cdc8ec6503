#!/usr/bin/env python3
"""``eval_lm`` -- the driver of the GNN+kNN eval path, mirror of ``fairseq_cli/eval_lm.py:61-336``
for ``--graph --use-precompute-feat [--knnlm]`` with the reference's flag names
(fairseq/options.py:472-501, fairseq/tasks/language_modeling.py:97-153,
fairseq/models/transformer_lm.py:122-139; recipe values: gnnlm_scripts/wiki103/hgt_lm_wiki103_reproduce.sh:81-89,140-150).

    python -m gnnlm_amd.eval_lm DATA --path CKPT --gen-subset test --graph --neighbor-context 2 --gcn-k 128 \\
        --use-precompute-feat --sample-break-mode none --max-tokens 256 --tokens-per-sample 256 \\
        --gcn-context-window 0 --knn-keytype gcn_feat \\
        [--knnlm --k 1024 --lmbda 0.25 --dstore-dir DATA/train_dstore --index-file ... --temperature 0.01 \\
         --knn-sim-func do_not_recomp_ip]

What is NOT rebuilt (out of scope): the fairseq task/dataset/checkpoint machinery.
The driver reads the data directory's own files instead: targets are ``{split}_dstore/vals.npy`` (the
same tokens in the same order, eval_lm.py:238-242), features ``{split}_dstore/keys.npy``, graph
neighbours ``neighbors.mmap.{gcn_k}``, PQ codes ``train_dstore/quantized-keys.npy``; blocks follow
``--sample-break-mode none`` (token_block_utils_fast.pyx:22-35).  Every table is uploaded to HBM once.
Output: the reference's two final lines (eval_lm.py:325-331).
"""
import argparse
import ast
import json
import struct
import logging
import math
import os
import sys
import time

import numpy as np
import torch

from . import ops
from .data_store import DataStore, vals_dtype
from .hgt import NeighborGraph
from .knn_model import KNNModel
from .model import GnnLmModel
from .path_utils import dstore_path, feature_path, neighbor_path, quantized_feature_path, value_path
from .sequence_scorer import SequenceScorer

logger = logging.getLogger("gnnlm_amd.eval_lm")


def get_parser():
    p = argparse.ArgumentParser("gnnlm-eval-lm")
    p.add_argument("data")
    p.add_argument("--path", required=True, help="checkpoint ({'args','model'} torch pickle)")
    p.add_argument("--gen-subset", default="test")
    p.add_argument("--task", default="language_modeling")
    p.add_argument("--graph", action="store_true", default=False)
    p.add_argument("--neighbor-context", default="(2, 2)")
    p.add_argument("--use-precompute-feat", action="store_true", default=False)
    p.add_argument("--invalid-neighbor-context", default=1536, type=int)
    p.add_argument("--gcn-k", default=1024, type=int)
    p.add_argument("--gcn-context-window", default=0, type=int)
    p.add_argument("--intra-context", default=0, type=int)
    p.add_argument("--sample-break-mode", default="none")
    p.add_argument("--tokens-per-sample", default=1024, type=int)
    p.add_argument("--max-tokens", default=None, type=int)
    p.add_argument("--max-sentences", default=None, type=int)
    p.add_argument("--batch-blocks", default=-1, type=int,
                   help="(this build) score this many of the batches' blocks per launch whatever --max-tokens says: blocks are "
                        "independent, so the hypotheses, their order and the scores are those of the one-block batches of the "
                        "recipe (`--max-tokens 256`), which alone are launch-bound on this part.  -1 (default): 32 when "
                        "--max-tokens gives one block per batch (128 with --knnlm: the on-device search likes big query blocks; 16 for models with "
                        "more than one HGT layer), else off; 0: off")
    p.add_argument("--softmax-batch", default=sys.maxsize, type=int)
    p.add_argument("--context-window", default=0, type=int)
    p.add_argument("--model-overrides", default="{}")
    p.add_argument("--knn-keytype", default=None)
    p.add_argument("--knnlm", action="store_true")
    p.add_argument("--k", default=1024, type=int)
    p.add_argument("--probe", default=8, type=int)
    p.add_argument("--lmbda", default=0.0, type=float)
    p.add_argument("--dstore-dir", default=None)
    p.add_argument("--index-file", default=None)
    p.add_argument("--temperature", default=1.0, type=float)
    p.add_argument("--knn-sim-func", default="do_not_recomp_ip")
    p.add_argument("--first", default=0, type=int)
    p.add_argument("--num-shards", default=1, type=int)
    p.add_argument("--shard-id", default=0, type=int)
    p.add_argument("--fp16", action="store_true")
    p.add_argument("--cpu", action="store_true")
    p.add_argument("--save-knnlm-dstore", action="store_true")
    p.add_argument("--dstore-mmap", default=None, help="where --save-knnlm-dstore writes <subset>_dstore[-keytype]/")
    p.add_argument("--dstore-fp16", action="store_true")
    p.add_argument("--output-word-probs", action="store_true")
    p.add_argument("--output-word-stats", action="store_true")
    p.add_argument("--output-knn-recall", action="store_true")
    p.add_argument("--remove-bpe", nargs="?", const="@@ ", default=None)
    p.add_argument("--add-bos-token", action="store_true", default=False)
    p.add_argument("--log-format", default=None)
    p.add_argument("--device", default="cuda:0", help="(this build) one process: the device; under torch.distributed.run every rank takes "
                                                      "cuda:LOCAL_RANK")
    p.add_argument("--store", choices=["auto", "replicated", "sharded", "peer"], default="auto",
                   help="(this build) multi-GPU runs (one process per GPU, `python -m torch.distributed.run --nproc-per-node N -m "
                        "gnnlm_amd.eval_lm ...`): where the PQ code table lives.  replicated: every rank holds all of it, no exchange; "
                        "sharded: by key range across the ranks (rank g owns rows [g*ceil(N/R), (g+1)*ceil(N/R)) plus a halo of the "
                        "context sizes), rows fetched from their owners by an RCCL all-to-all, equal context groups merged BEFORE the "
                        "exchange; peer: range shards mapped into every rank (HIP IPC), the kernels read rows from their owner's HBM "
                        "over xGMI themselves.  auto: sharded when WORLD_SIZE > 1, else the one table")
    p.add_argument("--graph-capture", action="store_true",
                   help="(this build) replay the model's forward and its softmax from HIP graphs captured once per batch shape and stream.  "
                        "Measured on the recipe's literal one-block batches: no gain -- the batch is bound by the DEVICE-side latency of its ~45 "
                        "dependent launches, not by the host that enqueues them (456 k tokens/s eager, 446 k replayed); what helps is --streams "
                        "(several batches in flight: 0.87 M tokens/s eager on 6 streams, 0.65 M replayed on 4).  Kept as an option; not with "
                        "--store sharded; multi-layer models are captured un-merged")
    p.add_argument("--streams", default=0, type=int,
                   help="(this build) score successive batches on this many HIP streams in turn (blocks are independent; every stream has its own "
                        "workspaces, graphs and score accumulator).  A one-block batch is a chain of ~45 small dependent launches -- 0.56 ms on the "
                        "device for 0.2 ms of work, whoever enqueues them -- and several chains in flight overlap.  Measured at the full store "
                        "(tokens/s, eight passes each): 1 stream 0.45 M; 3: 0.60 or 0.87 M and 4: 0.69 or 0.95 M depending on which hardware queues "
                        "the streams of a call land on; 6: 0.85-0.88 M every time; 8: 0.82-0.98 M.  0 (default): 6 when the batches are single blocks "
                        "(the recipe's `--max-tokens 256` with --batch-blocks 0), else 1.  Not with --store sharded "
                        "(collectives stay on one stream)")
    p.add_argument("--result-json", default=None,
                   help="(this build) write the run's figures (score_sum, count, ppl, tokens, seconds, per-rank sums, xGMI bytes) to this "
                        "file as JSON; in a multi-process run rank r writes PATH.rank<r> and rank 0 also PATH")
    p.add_argument("--exchange", choices=["exact", "padded"], default="exact",
                   help="(this build) --store sharded: variable-split exchange that never drops a row (default), or the fixed-capacity "
                        "sync-free one (checked at the end of the run)")
    return p


class _Dict:
    """pad/eos ids of a fairseq Dictionary (fairseq/data/dictionary.py: bos=0, pad=1, eos=2, unk=3)."""

    def pad(self):
        return 1

    def eos(self):
        return 2


class WordStat(object):
    """Per-word accumulator of --output-word-stats (fairseq_cli/eval_lm.py:32-58)."""

    def __init__(self, word, is_bpe):
        self.word, self.is_bpe = word, is_bpe
        self.log_prob, self.next_word_prob, self.count, self.missing_next_words = 0, 0, 0, 0

    def add(self, log_prob, next_word_prob):
        if next_word_prob is not None:
            self.next_word_prob += next_word_prob
        else:
            self.missing_next_words += 1
        self.log_prob += log_prob
        self.count += 1

    def __str__(self):
        return '{}\t{}\t{}\t{}\t{}\t{}'.format(self.word, self.count, self.log_prob, self.is_bpe,
                                               self.next_word_prob, self.count - self.missing_next_words)


def load_symbols(data):
    """Token id -> string of the data directory's fairseq dictionary (``dict.txt``: one ``<symbol> <count>`` line per entry
    after the four specials, fairseq/data/dictionary.py); None if the file is absent (words are then printed as ids)."""
    f = os.path.join(data, "dict.txt")
    if not os.path.exists(f):
        return None
    syms = ["<s>", "<pad>", "</s>", "<unk>"]
    with open(f, encoding="utf-8") as fh:
        for line in fh:
            line = line.rstrip("\n")
            if line:
                syms.append(line.rsplit(" ", 1)[0])
    return syms


def word_outputs(args, hypos, sample_ids, symbols, bpe_toks, bpe_len, word_stats):
    """--output-word-probs / --output-word-stats / --output-knn-recall (fairseq_cli/eval_lm.py:246-313), hypothesis by
    hypothesis on the host (an opt-in reporting path: it synchronises).  Returns the number of BPE continuation tokens
    whose scores were folded into the next token (they do not count as words, :258-263,274)."""
    skipped_total = 0
    for i, hyp in enumerate(hypos):
        h = hyp[0]
        tokens = h["tokens"].cpu()
        pos_scores = h["positional_scores"].float().cpu().clone()
        knn_recall = h["knn_recall"].cpu() if h["knn_recall"] is not None else None
        if args.add_bos_token:
            tokens, pos_scores = tokens[1:], pos_scores[1:]
        tgt_len = tokens.numel()
        if bpe_toks is not None:
            for t in range(tgt_len - 1):
                if tokens[t].item() in bpe_toks:
                    skipped_total += 1
                    pos_scores[t + 1] += pos_scores[t]
                    pos_scores[t] = 0
        w, word_prob, is_bpe = "", [], False
        for t in range(len(tokens)):
            w_ind = tokens[t].item()
            w += symbols[w_ind] if symbols is not None and w_ind < len(symbols) else str(w_ind)
            if bpe_toks is not None and w_ind in bpe_toks:
                w = w[:-bpe_len]
                is_bpe = True
            else:
                if args.output_knn_recall:
                    word_prob.append((w, pos_scores[t].item(), knn_recall[t] if knn_recall is not None else None))
                else:
                    word_prob.append((w, pos_scores[t].item()))
                next_prob, ind = None, t + 1
                while ind < len(tokens):
                    if pos_scores[ind].item() != 0:
                        next_prob = pos_scores[ind]
                        break
                    ind += 1
                word_stats.setdefault(w, WordStat(w, is_bpe)).add(pos_scores[t].item(), next_prob)
                is_bpe, w = False, ""
        if args.output_word_probs:
            if args.output_knn_recall:
                logger.info(str(int(sample_ids[i])) + " " + ('\t'.join('{} [{:2f}] [{}]'.format(x[0], x[1], x[2]) for x in word_prob)))
            else:
                logger.info(str(int(sample_ids[i])) + " " + ('\t'.join('{} [{:2f}]'.format(x[0], x[1]) for x in word_prob)))
    return skipped_total


def block_ranges(n_tokens, block, context_window=0):
    """(context_start, start, end) of every block under --sample-break-mode none
    (token_block_utils_fast.pyx:22-35) with the --gcn-context-window prefix of
    GraphTokenBlockDataset.get_basic_info (token_block_dataset.py:246-285)."""
    out = []
    for s in range(0, n_tokens, block):
        e = min(n_tokens, s + block)
        out.append((max(0, s - context_window) if s > 0 else s, s, e))
    return out


_MMAP_IDX_DTYPES = {1: np.uint8, 2: np.int8, 3: np.int16, 4: np.int32, 5: np.int64, 6: np.float64, 7: np.float64, 8: np.uint16}


def fairseq_token_stream(data, split):
    """The split's token stream as fairseq binarised it (``DATA/{split}.bin/.idx``, MMapIndexedDataset:
    fairseq/data/indexed_dataset.py:350-420): header ``MMIDIDX\0\0`` + <Q version 1 + <B dtype code + <Q n_sentences, then
    int32 sizes.  Read without fairseq; None when the pair is absent or in another format (the legacy ``TNTIDX`` index)."""
    idx, binf = os.path.join(data, split + ".idx"), os.path.join(data, split + ".bin")
    if not (os.path.exists(idx) and os.path.exists(binf)):
        return None
    with open(idx, "rb") as f:
        if f.read(9) != b"MMIDIDX\x00\x00":
            return None
        version, code, n_sent = struct.unpack("<QBQ", f.read(17))
        if version != 1 or code not in _MMAP_IDX_DTYPES:
            return None
        sizes = np.frombuffer(f.read(4 * n_sent), dtype=np.int32)
    total = int(sizes.astype(np.int64).sum())
    return np.memmap(binf, mode="r", dtype=_MMAP_IDX_DTYPES[code], shape=(total,))


def check_tables(args, info, n_store):
    """The driver takes the targets from ``{split}_dstore/vals.npy`` (row i = the i-th token of the split: the reference saves
    them in iteration order, fairseq_cli/eval_lm.py:238-242) instead of fairseq's ``.bin/.idx``.  That is only the same thing
    when every per-token file of the split describes the same tokens -- checked here instead of assumed: file sizes against
    ``info.json``, and the token stream itself whenever the binarised split is there."""
    split, data = args.gen_subset, args.data
    n_tok, d = info["dstore_size"], info["hidden_size"]
    vdt = np.dtype(vals_dtype(info["dstore_fp16"], info.get("vocab_size")))
    if info.get("val_size", 1) != 1:
        raise ValueError(f"{dstore_path(data, split)}/info.json: val_size must be 1 (labels), found {info.get('val_size')}")
    want = {feature_path(data, split): n_tok * d * (2 if info["dstore_fp16"] else 4),
            value_path(data, split): n_tok * vdt.itemsize,
            neighbor_path(data, split, args.gcn_k): n_tok * args.gcn_k * 8}
    for path, size in want.items():
        if not os.path.exists(path):
            raise FileNotFoundError("Dataset not found: {} ({})".format(split, path))
        have = os.path.getsize(path)
        if have != size:
            raise ValueError(f"{path}: {have} bytes, but info.json (dstore_size {n_tok}, hidden_size {d}, fp16 {info['dstore_fp16']}, "
                             f"vocab {info.get('vocab_size')}) and --gcn-k {args.gcn_k} imply {size}: the per-token files of the "
                             f"'{split}' split do not describe the same tokens")
    stream = fairseq_token_stream(data, split)
    if stream is not None:
        if stream.shape[0] < n_tok:
            raise ValueError(f"{data}/{split}.bin holds {stream.shape[0]} tokens, {dstore_path(data, split)} has {n_tok} rows")
        vals = np.memmap(value_path(data, split), mode="r", shape=(n_tok,), dtype=vdt)
        for s in range(0, n_tok, 1 << 24):                      # targets = the token stream itself (monolingual_dataset.py:86-94)
            a, b = np.asarray(stream[s:s + (1 << 24)]).astype(np.int64), np.asarray(vals[s:s + (1 << 24)]).astype(np.int64)
            if not np.array_equal(a[:len(b)], b):
                at = s + int(np.nonzero(a[:len(b)] != b)[0][0])
                raise ValueError(f"{value_path(data, split)} row {at} = {int(vals[at])} but token {at} of {data}/{split}.bin is "
                                 f"{int(stream[at])}: the datastore was not written over this split in corpus order")
        logger.info("targets of %s_dstore/vals.npy == token stream of %s.bin (%d tokens)", split, split, n_tok)


def load_tables(args, device, shard=None):
    """``shard`` (dist.Shard): upload only the rows of the code table this rank holds (its key range plus the halo)."""
    split, data = args.gen_subset, args.data
    info = json.load(open(os.path.join(dstore_path(data, split), "info.json")))
    tinfo = json.load(open(os.path.join(dstore_path(data, "train"), "info.json")))        # language_modeling.py:266-272
    n_tok, d = info["dstore_size"], info["hidden_size"]
    if not os.path.exists(feature_path(data, split)):
        raise FileNotFoundError("Dataset not found: {} ({})".format(split, feature_path(data, split)))
    check_tables(args, info, tinfo["dstore_size"])
    feats = np.memmap(feature_path(data, split), mode="r", shape=(n_tok, d),
                      dtype=np.float16 if info["dstore_fp16"] else np.float32)
    targets = np.memmap(value_path(data, split), mode="r", shape=(n_tok,),
                        dtype=vals_dtype(info["dstore_fp16"], info.get("vocab_size")))
    nbrs = np.memmap(neighbor_path(data, split, args.gcn_k), mode="r", dtype=np.int64, shape=(n_tok, args.gcn_k))
    codes = np.load(quantized_feature_path(data, "train"), mmap_mode="r")                  # language_modeling.py:274-276
    if codes.ndim != 2 or codes.dtype != np.uint8 or codes.shape[0] != tinfo["dstore_size"]:
        raise ValueError(f"{quantized_feature_path(data, 'train')}: {codes.dtype} {codes.shape}, expected uint8 "
                         f"[{tinfo['dstore_size']}, M] (one code row per key of train_dstore)")
    up = lambda a: torch.from_numpy(np.array(a)).to(device)
    tg = up(targets).long()
    if shard is not None:
        if shard.n_store != tinfo["dstore_size"]:
            raise ValueError("the shard was laid out for another store size")
        codes = codes[shard.store_row0:shard.store_row0 + shard.store_rows]
    return {"n_tok": n_tok, "d": d, "vocab": info.get("vocab_size"), "n_store": tinfo["dstore_size"],
            "feats": up(feats), "targets": tg, "nbrs": up(nbrs), "codes": up(codes),
            "codes_row0": shard.store_row0 if shard is not None else 0,
            "no_pad": not bool(tg.eq(_Dict().pad()).any())}           # one look at the split instead of one sync per hypothesis


def main(args, tables=None, model=None):
    """``tables`` / ``model``: already-resident tables (the dict of :func:`load_tables`) and a built
    :class:`GnnLmModel` -- bench.py times this driver on the 103 M-row store it generated on the device instead of
    writing 13 GB of ``quantized-keys.npy`` first.  Everything else is the reference's loop."""
    if args.cpu or not torch.cuda.is_available():
        raise RuntimeError("gnnlm_amd.eval_lm needs an MI355X: the product path has no CPU fallback")
    if not (args.graph and args.use_precompute_feat):
        raise NotImplementedError("only the --graph --use-precompute-feat eval path is built (the base LM is out of scope)")
    if args.save_knnlm_dstore and not args.dstore_mmap:
        raise ValueError("--save-knnlm-dstore needs --dstore-mmap")
    if args.knnlm and args.save_knnlm_dstore:
        raise ValueError("Cannot use knnlm while trying to build the datastore!")
    if args.context_window > 0:
        # LMContextWindowDataset (fairseq/data/lm_context_window_dataset.py) prepends context tokens and scores only the
        # new ones; shrinking the block without the prefix would silently give another ppl
        raise NotImplementedError("--context-window > 0 is not built (the GNN-LM recipes use --gcn-context-window)")
    if args.fp16:
        logger.warning("--fp16 ignored: the HIP path computes in float32")
    # ---- one process per GPU (torch.distributed.run): token blocks are data parallel, the code table is range-sharded
    # (the reference has no multi-GPU eval path to mirror beyond --num-shards, fairseq_cli/eval_lm.py:131-132)
    dist = torch.distributed
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    own_group = False
    if world > 1:
        device = torch.device("cuda", int(os.environ.get("GNNLM_EVAL_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
        torch.cuda.set_device(device)
        if not dist.is_initialized():
            backend = os.environ.get("GNNLM_EVAL_BACKEND", "nccl")                          # "gloo": tests, several ranks on one GPU
            if backend != "nccl":
                os.environ["GNNLM_TEST_HOST_STAGED"] = "1"
                dist.init_process_group(backend, rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            own_group = True
        if args.save_knnlm_dstore:
            raise NotImplementedError("--save-knnlm-dstore writes ONE datastore: run it with one process")
    else:
        device = torch.device(args.device)
        torch.cuda.set_device(device)
    store_mode = args.store if args.store != "auto" else ("sharded" if world > 1 else "replicated")
    if world == 1 and store_mode != "replicated" and not os.environ.get("GNNLM_EVAL_FORCE_EXCHANGE"):
        store_mode = "replicated"                                                           # one rank owns every row
    nc = ast.literal_eval(str(args.neighbor_context))                                      # language_modeling.py:295
    left, right = (nc, nc) if isinstance(nc, int) else nc
    shard = None
    if store_mode != "replicated":
        from .dist import Shard
        n_store = tables["n_store"] if tables is not None else \
            json.load(open(os.path.join(dstore_path(args.data, "train"), "info.json")))["dstore_size"]
        shard = Shard(n_store, world, rank, halo_left=left, halo_right=right)
    tabs = tables if tables is not None else load_tables(args, device, shard)
    overrides = ast.literal_eval(args.model_overrides)
    if model is None:
        model, margs = GnnLmModel.from_checkpoint(args.path, device, overrides, vocab_size=tabs["vocab"])
    fetcher = None
    if shard is None:
        store = model.make_store(tabs["codes"], tabs["n_store"], device)
    else:
        from .dist import PeerMappedFetcher, ShardedFetcher
        codes = tabs["codes"]
        if codes.shape[0] != shard.store_rows:                                             # (resident tables handed in whole: cut this rank's rows)
            codes = codes[shard.store_row0:shard.store_row0 + shard.store_rows].contiguous()
        store = model.make_store(codes, tabs["n_store"], device, row0=shard.store_row0)
        if store_mode == "peer":
            store = PeerMappedFetcher(store, shard).mapped_store()                          # the kernels read every row from its owner themselves
        else:
            fetcher = ShardedFetcher(store, shard, mode=args.exchange)
    T = args.tokens_per_sample - args.context_window
    blocks = [r + (bid,) for bid, r in enumerate(block_ranges(tabs["n_tok"], T, args.gcn_context_window))]   # (context start, start, end, sample id)
    if args.first > 0:
        blocks = blocks[:args.first]
    blocks = blocks[args.shard_id::args.num_shards]                                        # eval_lm.py:131-132
    if world > 1:                                                                          # this rank's contiguous share of the blocks
        per_rank = -(-len(blocks) // world)
        blocks = blocks[rank * per_rank:(rank + 1) * per_rank]
    per_batch = max(1, (args.max_tokens or 36000) // max(1, T + args.gcn_context_window))
    if args.max_sentences:
        per_batch = min(per_batch, args.max_sentences)
    if args.batch_blocks < 0:                       # auto: only the one-block batches of the recipe are coalesced -- with B > 1 per
        # batch the reference's scorer has its own target / query pairing.  32 blocks at one HGT layer (the step is launch-bound
        # below that); deeper models carry (1 + l + r) k_g rows of state per token and layer and fill the chip from 4 blocks on --
        # 16 because equal context groups of a batch are computed once and more blocks share more of them (53 GB of workspace
        # at the recipe's shapes if nothing merges; measured: 16.1 k tokens/s against 15.9 k at 4 blocks on i.i.d. ids, 39.8 k
        # against 35.1 k on searched neighbours)
        deep = getattr(getattr(model, "hgt_decoder", None), "n_layers", 1) > 1
        # with the kNN search on the device (--knnlm) 128: 32768 queries per search fill its 8-query groups and keep a list's bytes in L2
        # (11.0 ms per 8192 queries against 12.2 in batches of 32 blocks; ~18 GB of search temporaries per batch in flight)
        args.batch_blocks = (16 if deep else (128 if args.knnlm else 32)) if per_batch == 1 else 0
    per_batch = max(per_batch, args.batch_blocks)
    # neighbours inside the token's own context are dropped on the TRAIN split only (language_modeling.py:299,
    # token_block_dataset.py:360-362): the split whose GNN features the kNN index is built over (find_knn.sh:7)
    invalid_ctx = args.invalid_neighbor_context if args.gen_subset == "train" else 0
    if getattr(args, "graph_capture", False) and fetcher is not None:
        raise ValueError("--graph-capture needs the one-table or the peer-mapped store (the exchange of --store sharded synchronises)")
    model.graph_capture = bool(getattr(args, "graph_capture", False))
    scorer = SequenceScorer(_Dict(), args.softmax_batch, args=args)
    knn_dstore = None
    if args.knnlm:
        knn_dstore = KNNModel(index_file=args.index_file, dstore_dir=args.dstore_dir, cuda=-1, k=args.k,
                              no_load_keys=("do_not_recomp" in args.knn_sim_func), use_memory=True,
                              metric_type=args.knn_sim_func, device=device) if not getattr(args, "knn_model", None) \
            else args.knn_model
    # --save-knnlm-dstore (fairseq_cli/eval_lm.py:103-104,178-205,222-242): the keys the scorer hands back per hypothesis
    # (`--knn-keytype`, here the HGT output "gcn_feat") and the target tokens, written as the split's datastore
    save = None
    if args.save_knnlm_dstore:
        dstore_size = int(tabs["n_tok"])                                   # dataset.sizes.sum() (:104)
        # key dimension = what the scorer hands back for --knn-keytype: the HGT output ("gcn_feat", out_dim of the output
        # adapter if there is one) or the precomputed features (the inner_states[-1] fallback, sequence_scorer.py:105)
        hd = getattr(model, "hgt_decoder", None)
        dim = int(getattr(hd, "out_dim", tabs["d"])) if (args.knn_keytype == "gcn_feat" and not getattr(model, "short_cut", False)) \
            else int(tabs["d"])
        fp16 = bool(args.dstore_fp16)
        suffix = "" if not args.knn_keytype else f"-{args.knn_keytype}"
        save_dir = os.path.join(args.dstore_mmap, f"{args.gen_subset}_dstore{suffix}")
        os.makedirs(save_dir, exist_ok=True)
        vocab = tabs.get("vocab") or int(tabs["targets"].max().item()) + 1
        info = {"dstore_size": dstore_size, "hidden_size": dim, "vocab_size": int(vocab), "dstore_fp16": fp16, "val_size": 1}
        logger.info(f"keytype being saved: {args.knn_keytype}")
        logger.info(f"dstore info: {info}")
        json.dump(info, open(os.path.join(save_dir, "info.json"), "w"), indent=4, sort_keys=True)
        save = {"dir": save_dir, "idx": 0, "size": dstore_size, "dim": dim,
                "keys": np.memmap(os.path.join(save_dir, "keys.npy"), dtype=np.float16 if fp16 else np.float32, mode="w+",
                                  shape=(dstore_size, dim)),
                "vals": np.memmap(os.path.join(save_dir, "vals.npy"), dtype=np.int16 if fp16 and vocab < 2 ** 15 else np.int32,
                                  mode="w+", shape=(dstore_size, 1))}
    # --output-word-probs / --output-word-stats / --output-knn-recall / --remove-bpe (fairseq_cli/eval_lm.py:146-160,246-313,333-336)
    want_words = args.output_word_probs or args.output_word_stats
    symbols, bpe_toks, bpe_len, word_stats = None, None, 0, dict()
    if want_words or args.remove_bpe is not None:
        symbols = load_symbols(args.data)
    if args.remove_bpe is not None:
        if args.remove_bpe == "sentencepiece":
            raise NotImplementedError                                        # as the reference (:148-149)
        if symbols is None:
            raise ValueError("--remove-bpe needs the data directory's dict.txt (the continuation marker lives in the symbols)")
        bpe_cont = args.remove_bpe.rstrip()
        bpe_toks = {i for i in range(len(symbols)) if symbols[i].endswith(bpe_cont)}
        bpe_len = len(bpe_cont)
    if args.output_knn_recall and not args.knnlm:
        raise ValueError("--output-knn-recall needs --knnlm (the scorer returns no recall otherwise)")
    acc = torch.zeros(1, device=device, dtype=torch.float64)
    hyp_sums = []
    count, ntok = 0, 0
    timers = []             # gen_timer (eval_lm.py:214-219) as HIP event pairs on the stream: no per-batch host sync
    # the batches of this rank: runs of equally long blocks, up to per_batch of them
    batches_, i = [], 0
    while i < len(blocks):
        group = [blocks[i]]
        while len(group) < per_batch and i + len(group) < len(blocks) and \
                (blocks[i + len(group)][2] - blocks[i + len(group)][0]) == (group[0][2] - group[0][0]):
            group.append(blocks[i + len(group)])
        i += len(group)
        batches_.append(group)
    idle_steps = 0
    if fetcher is not None and world > 1:
        # the exchange is a collective: every rank takes part in as many of them as the rank with the most batches
        t = torch.tensor([len(batches_)], dtype=torch.int64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        idle_steps = int(t.item()) - len(batches_)
        if args.exchange == "padded":
            # equal-split all-to-alls: every rank sizes its buckets from the SAME request count, the largest batch of any rank
            t = torch.tensor([max((len(g_) * (g_[0][2] - g_[0][0]) for g_ in batches_), default=0) * args.gcn_k], dtype=torch.int64, device=t.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            fetcher.fixed_requests = int(t.item())
    deep = getattr(getattr(model, "hgt_decoder", None), "n_layers", 1) > 1
    # lanes: 6 for the recipe's literal one-block batches (launch-bound); 2 for coalesced batches scored WITH a device-side kNN search
    # (the host comes back to a batch's search when the next batch is enqueued: generate_begin / generate_finish); else 1
    pipelined = bool(args.knnlm) and hasattr(scorer, "generate_begin") and hasattr(knn_dstore, "interpolate_begin") and fetcher is None and save is None
    n_streams = args.streams if getattr(args, "streams", 0) > 0 else \
        (6 if per_batch == 1 and fetcher is None and save is None else (2 if pipelined and per_batch > 1 and not deep else 1))   # (a multi-layer model's centre-state cache and its 3-GB-per-block workspace belong to ONE stream)
    pipelined = pipelined and n_streams > 1
    if n_streams > 1 and fetcher is not None:
        raise ValueError("--streams > 1 is not available with --store sharded (the exchange's collectives stay on one stream)")
    main_stream = torch.cuda.current_stream(device)
    lanes = [main_stream] + [torch.cuda.Stream(device=device) for _ in range(n_streams - 1)]
    accs = [acc] + [torch.zeros_like(acc) for _ in lanes[1:]]
    for s_ in lanes[1:]:
        s_.wait_stream(main_stream)
    in_flight, order = [None] * n_streams, [0] * n_streams
    state = {"ntok": 0, "count": 0}

    def consume(hypos, sample, acc, ev0, ev1):
        """What the loop does with a batch's hypotheses (a handle of generate_begin is finished first), on the batch's own stream."""
        if isinstance(hypos, dict):
            hypos = scorer.generate_finish(hypos)
        ev1.record()
        timers.append((ev0, ev1))
        state["ntok"] += sample["ntokens"]
        if save is not None:                                     # one device -> host copy per batch (this run is a writer anyway)
            kd = hypos[0][0]["dstore_keys"].shape[-1]
            if kd != save["dim"]:
                raise ValueError(f"--save-knnlm-dstore: the scorer returned {kd}-dimensional keys, the datastore was opened for {save['dim']}")
            keys = torch.cat([h[0]["dstore_keys"].reshape(-1, kd) for h in hypos])
            toks = torch.cat([h[0]["tokens"].reshape(-1) for h in hypos])
            n_new = min(keys.shape[0], save["size"] - save["idx"])
            if n_new < keys.shape[0]:
                logger.warning("exceed offset at sample " + str(int(sample["id"][0])))          # :227-230
            sl = slice(save["idx"], save["idx"] + n_new)
            save["keys"][sl] = keys[:n_new].to(torch.float16 if save["keys"].dtype == np.float16 else torch.float32).cpu().numpy()
            save["vals"][sl, 0] = toks[:n_new].cpu().numpy().astype(save["vals"].dtype)
            save["idx"] += n_new
        pos = torch.cat([h[0]["positional_scores"].float().reshape(-1) for h in hypos])     # one launch per batch
        ops.masked_sum_f64(pos, None, acc)                                                  # score_sum (:273), in f64
        # ... and as the reference adds it up: one float32 sum per hypothesis (`pos_scores.sum()`), accumulated in a float32 scalar
        # (`score_sum += ...cpu()`, :273) -- the per-hypothesis sums are kept on the device and chained on the host at the end
        lens = [h[0]["positional_scores"].numel() for h in hypos]
        hyp_sums.append(pos.view(len(hypos), -1).sum(dim=1) if len(set(lens)) == 1 and lens[0] > 0 else
                        torch.stack([h[0]["positional_scores"].float().sum() for h in hypos]))
        state["count"] += pos.numel()                                                                # :274
        if want_words or bpe_toks is not None:
            state["count"] -= word_outputs(args, hypos, sample["id"], symbols, bpe_toks, bpe_len, word_stats)     # skipped_toks (:274)

    torch.cuda.synchronize()
    wall0 = time.perf_counter()
    for bi, group in enumerate(batches_):
        if n_streams > 1:                                       # this batch's lane: its stream, its accumulator
            torch.cuda.set_stream(lanes[bi % n_streams])
            acc = accs[bi % n_streams]
        L = group[0][2] - group[0][0]
        if all(group[j + 1][0] == group[j][2] for j in range(len(group) - 1)):
            idx = slice(group[0][0], group[-1][2])           # back-to-back blocks (no --gcn-context-window): plain views, no gather
        else:
            idx = torch.cat([torch.arange(g_[0], g_[2], device=device) for g_ in group])
        target = tabs["targets"][idx].view(len(group), L)
        nb_ids = tabs["nbrs"][idx].contiguous()
        if invalid_ctx > 0:
            tok_pos = torch.arange(idx.start, idx.stop, device=device) if isinstance(idx, slice) else idx     # global offsets in the split
            nb_ids = ops.filter_neighbors(nb_ids, tok_pos.contiguous(), invalid_ctx)
        graph = NeighborGraph(ids=nb_ids, n_blocks=len(group), T=L, left=left, right=right,
                              store=store, tgt_h=tabs["feats"][idx].contiguous(), max_intra_context=args.intra_context, fetcher=fetcher)
        sample = {"id": torch.tensor([g_[3] for g_ in group]), "nsentences": len(group), "ntokens": len(group) * L,
                  "net_input": {"src_tokens": target, "src_lengths": torch.full((len(group),), L), "graph": graph},
                  "target": target, "start_indices": [g_[1] - g_[0] for g_ in group]}
        if tabs.get("no_pad") is not None:
            sample["no_pad_in_target"] = tabs["no_pad"]
        if args.batch_blocks > 0:
            sample["blockwise_knn"] = True                       # kNN pairing of one-block batches (sequence_scorer.py)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if pipelined:
            lane = bi % n_streams
            if in_flight[lane] is not None:
                consume(*in_flight[lane])                                   # (this lane's previous batch: its search has had a whole batch of head start)
            ev0.record()
            in_flight[lane] = (scorer.generate_begin([model], sample, knn_dstore=knn_dstore, temperature=args.temperature), sample, acc, ev0, ev1)
            order[lane] = bi
            continue
        ev0.record()
        hypos = scorer.generate([model], sample, knn_dstore=knn_dstore, temperature=args.temperature) if args.knnlm \
            else scorer.generate([model], sample)
        consume(hypos, sample, acc, ev0, ev1)
    for lane in sorted((l_ for l_ in range(n_streams) if in_flight[l_] is not None), key=lambda l_: order[l_]):   # what is still in flight, in batch order
        torch.cuda.set_stream(lanes[lane])
        consume(*in_flight[lane])
        in_flight[lane] = None
    ntok, count = ntok + state["ntok"], count + state["count"]
    if n_streams > 1:
        torch.cuda.set_stream(main_stream)
        for s_, a_ in zip(lanes[1:], accs[1:]):
            main_stream.wait_stream(s_)
            accs[0] += a_
        acc = accs[0]
        # what the model keeps per stream (merge tables of n_store x 4 bytes, scratch arenas) goes with the side lanes: the next
        # call makes new streams, and torch's pool hands out up to 32 different handles
        # (the model drops the HIP graphs it captured for those lanes with them: their launches have the workspace addresses baked in)
        if hasattr(model, "release_stream_state"):
            model.release_stream_state(keep=(main_stream.cuda_stream,))
    for _ in range(idle_steps):                                                             # no batch left here: serve the peers' requests
        if deep and getattr(model.hgt_decoder, "dedup_groups", False):
            fetcher.fetch_groups(torch.empty(0, dtype=torch.int64, device=device), left, right, torch.zeros(4, dtype=torch.int32, device=device))
        else:
            fetcher.fetch_codes(torch.empty(0, args.gcn_k, dtype=torch.int64, device=device), left, right, not deep)
    if fetcher is not None:
        fetcher.check()                                                                     # (padded exchange: nothing was dropped)
    score_sum = acc.item()                                                                  # the only host sync
    if save is not None:
        save["keys"].flush()
        save["vals"].flush()
        logger.info(f"Saved {save['idx']} data to {save['dir']}")                           # :322-323
    wall = time.perf_counter() - wall0                                                      # the loop as a whole ("wps", :316)
    gen_time = sum(a.elapsed_time(b) for a, b in timers) / 1e3
    if n_streams > 1:
        gen_time = min(gen_time, wall)                                                      # (the batches' intervals overlap: their sum is not a duration)
    rank_score_sum, rank_tokens = score_sum, ntok
    score_sum_f32 = np.float32(0.0)
    if hyp_sums:
        for v in torch.cat(hyp_sums).cpu().numpy():                                         # float32 + float32, hypothesis by hypothesis
            score_sum_f32 = np.float32(score_sum_f32 + v)
    count_rank = count
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        # the one reduction of the data-parallel run (16 B + the timer): score_sum, count, tokens summed; the time is the slowest rank's
        host = dist.get_backend() != "nccl"
        t = torch.tensor([score_sum, float(count), float(ntok)], device="cpu" if host else device, dtype=torch.float64)
        dist.all_reduce(t)
        tm = torch.tensor([gen_time, wall], device="cpu" if host else device, dtype=torch.float64)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        score_sum, count, ntok = t[0].item(), int(t[1].item()), int(t[2].item())
        gen_time, wall = tm[0].item(), tm[1].item()
    avg_nll_loss = -score_sum / count / math.log(2)
    line1 = "Evaluated {} tokens in {:.1f}s ({:.2f} tokens/s)".format(ntok, gen_time, ntok / max(gen_time, 1e-9))
    line2 = "Loss (base 2): {:.4f}, Perplexity: {:.2f}".format(avg_nll_loss, 2 ** avg_nll_loss)
    if rank == 0 or not (dist.is_available() and dist.is_initialized()):
        logger.info(line1)
        logger.info(line2)
        print(line1)
        print(line2)
    # the same with the reference's float32 accumulation (fairseq_cli/eval_lm.py:273-274): logged, not printed --
    # the two lines above are the reference's output.  Per rank in a multi-process run (the reference has one accumulator).
    if count_rank:
        nll32 = -float(score_sum_f32) / count_rank / math.log(2)
        logger.info("float32 accumulation order (as the reference, rank {}): Loss (base 2): {:.4f}, Perplexity: {:.2f}".format(rank, nll32, 2 ** nll32))
    if args.output_word_stats:                                                              # :333-336
        for ws in sorted(word_stats.values(), key=lambda x: x.count, reverse=True):
            logger.info(ws)
    link_bytes = getattr(fetcher, "link_bytes", None)
    if own_group:
        dist.destroy_process_group()
    res = {"score_sum": score_sum, "count": count, "ppl": 2 ** avg_nll_loss, "tokens": ntok, "seconds": gen_time,
            "wall_seconds": wall, "word_stats": word_stats if args.output_word_stats else None,
            "score_sum_f32_order": float(score_sum_f32), "rank": rank, "world": world, "store": store_mode, "rank_score_sum": rank_score_sum, "rank_tokens": rank_tokens,
            "xgmi_bytes": link_bytes}
    if getattr(args, "result_json", None):
        js = {k_: v for k_, v in res.items() if k_ != "word_stats"}
        for path in ([args.result_json + f".rank{rank}"] if world > 1 else []) + ([args.result_json] if rank == 0 else []):
            with open(path, "w") as fh:
                json.dump(js, fh)
    return res


def cli_main(argv=None):
    logging.basicConfig(level=logging.INFO, stream=sys.stderr)
    return main(get_parser().parse_args(argv))


if __name__ == "__main__":
    cli_main()
