#!/usr/bin/env python3
"""cProfile of the drop-in driver on the recipe's LITERAL one-block batches (`--max-tokens 256 --batch-blocks 0`): where the
host spends the 0.55 ms a 256-token batch takes (the GPU work of that batch is ~0.2 ms)."""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

sys.argv = [sys.argv[0], "--no-cpu-baseline", "--no-parity", "--no-extras", "--n-store", os.environ.get("N_STORE", "4000000")]
args = bench.parse()
dev = torch.device("cuda:0")
eng, shard, sharded, cpu_model, (d, vocab) = bench.build(args, dev, 0, 1)
batches = bench.make_batches(args, dev, 0, d, vocab)
import contextlib
from gnnlm_amd import eval_lm, ops
from gnnlm_amd.model import GnnLmModel
st, T = eng.store, args.tokens_per_sample
cat = lambda a: torch.cat([getattr(b, a) for b in batches])
n = cat("targets").shape[0]
sims, kids = cat("knn_sims"), cat("knn_ids")
model = GnnLmModel(eng.hgt, eng.asm, None)
model.make_store = lambda codes, n_store, device: st


class Knn:
    pos = 0

    def interpolate(self, queries, targets, lm_logp, t, lmbda, k=0):
        m = queries.shape[0]
        sl = slice(self.pos, self.pos + m)
        self.pos = (self.pos + m) % n
        return ops.knn_interp(lm_logp.contiguous(), sims[sl].contiguous(), kids[sl].contiguous(), targets.long().contiguous(), t, lmbda, vals=st.vals, n_store=st.n_store)


tabs = {"n_tok": n, "d": d, "vocab": None, "n_store": st.n_store, "feats": cat("tgt_feats"), "targets": cat("targets").clamp(min=4), "nbrs": cat("ids"),
        "codes": st.codes, "no_pad": True}
VARIANTS = [[], ["--graph-capture"]] if not os.environ.get("SWEEP") else \
    ([["--streams", os.environ.get("NS", "3")]] * 8 if os.environ.get("SWEEP") == "3" else [["--streams", "1"], ["--streams", "2"], ["--streams", "3"], ["--streams", "4"], ["--streams", "6"], ["--streams", "8"], ["--graph-capture", "--streams", "1"], ["--graph-capture", "--streams", "4"]] * 2)
for extra in VARIANTS:
    a = eval_lm.get_parser().parse_args(extra + ["-", "--path", "-", "--graph", "--use-precompute-feat", "--neighbor-context", "2", "--gcn-k", str(args.gcn_k),
                                                 "--tokens-per-sample", str(T), "--max-tokens", str(T), "--knnlm", "--k", str(args.k), "--lmbda", "0.25", "--temperature", "0.01",
                                                 "--knn-keytype", "gcn_feat", "--softmax-batch", str(64 * T), "--device", str(dev), "--batch-blocks", "0"])
    a.knn_model = Knn()
    with contextlib.redirect_stdout(io.StringIO()):
        eval_lm.main(a, tables=tabs, model=model)
    pr = cProfile.Profile()
    with contextlib.redirect_stdout(io.StringIO()):
        if not os.environ.get("SWEEP"):
            pr.enable()
        r = eval_lm.main(a, tables=tabs, model=model)
        pr.disable()
    print(extra, "tokens/s wall", round(r["tokens"] / r["wall_seconds"]), "batches", r["tokens"] // T, "us per batch", round(r["wall_seconds"] / (r["tokens"] // T) * 1e6, 1))
    if os.environ.get("SWEEP"):
        continue
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[4:40]))
