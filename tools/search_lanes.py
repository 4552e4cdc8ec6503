#!/usr/bin/env python3
"""The with-search step (HGT features -> on-device IVF-PQ search of those features -> adaptive softmax -> kNN
interpolation: what the reference's timer spans, fairseq_cli/eval_lm.py:214-219) with several batches in flight.

LANES batches are in flight at once (a lane's previous batch is finished -- the search's one host read, the
interpolation -- right before the lane's next batch is enqueued); lane j runs on stream j % STREAMS.
(LANES, STREAMS) = (1, 1): the serial step; (2, 1): the host's round trip hidden, one stream; (2, 2), (3, 3): independent
chains on separate streams (the latency-bound kernels of one batch's search under the other batch's GEMMs / filter).
Prints ms per step (8192 tokens) for every configuration in CONFIGS ("1x1,2x1,2x2,3x3")."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from gnnlm_amd import ops
from gnnlm_amd.synthetic import synthetic_ivfpq_index


def main():
    configs = [tuple(int(x) for x in c.split("x")) for c in os.environ.get("CONFIGS", "1x1,2x1,2x2,3x3,4x2").split(",")]
    steps = int(os.environ.get("STEPS", 24))
    sys.argv = [sys.argv[0], "--pool", "4"] + sys.argv[1:]
    args = bench.parse()
    dev = torch.device("cuda:0")
    eng, shard, sharded, cpu_model, (d, vocab) = bench.build(args, dev, 0, 1)
    batches = bench.make_batches(args, dev, 0, d, vocab)
    idx = synthetic_ivfpq_index(args.n_store, eng.hgt.hidden_dim, 4096, 64, dev, nprobe=32)
    idx.attach_vals(eng.store.vals)
    st = eng.store

    def begin(b):
        x = eng.features(b)
        qn = x / (x ** 2).sum(-1, keepdim=True).sqrt()
        h = idx.search_begin(qn.contiguous(), args.k, return_vals=True)
        lm = eng.asm.target_log_prob(x, b.targets)
        return h, lm, b

    def finish(p, acc):
        h, lm, b = p
        sims, ids, vals = h.result()
        logp, _, _ = ops.knn_interp(lm, sims, ids, b.targets, args.temperature, args.lmbda, n_store=st.n_store, knn_vals=vals)
        ops.masked_sum_f64(logp, None, acc)

    main_s = torch.cuda.current_stream()
    ref = None
    for lanes, streams in configs:
        ss = [main_s] + [torch.cuda.Stream(device=dev) for _ in range(streams - 1)]
        accs = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(lanes)]

        def run(n):
            pend = [None] * lanes
            for i in range(n):
                j = i % lanes
                with torch.cuda.stream(ss[j % streams]):
                    if pend[j] is not None:
                        finish(pend[j], accs[j])
                    pend[j] = begin(batches[i % len(batches)])
            for j in range(lanes):
                if pend[j] is not None:
                    with torch.cuda.stream(ss[j % streams]):
                        finish(pend[j], accs[j])
            torch.cuda.synchronize()
        for s_ in ss[1:]:
            s_.wait_stream(main_s)
        run(2 * lanes)                                            # allocations of every lane
        for a in accs:
            a.zero_()
        times = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(steps)
            times.append((time.perf_counter() - t0) / steps * 1e3)
        tot = sum(a.item() for a in accs) / 3
        if ref is None:
            ref = tot
        n_tok = batches[0].targets.shape[0]
        print(f"lanes {lanes} streams {streams}: {sorted(times)[1]:.3f} ms per step ({[round(t, 3) for t in times]}) = "
              f"{n_tok / sorted(times)[1] * 1e3:.0f} tokens/s; score sum {tot:.6f} (first config {ref:.6f})", flush=True)


if __name__ == "__main__":
    main()
