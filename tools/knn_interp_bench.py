#!/usr/bin/env python3
"""knn_interp with ids-only search results (the label gather of knn/knn_model.py:198) at the bench's shape: 8192 tokens x k = 1024
over a 103,227,021-row int32 label table -- the plain 4-byte gather against the one-byte tag table, and the labels-delivered path.
    python tools/knn_interp_bench.py [--n 8192] [--k 1024] [--rows 103227021]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gnnlm_amd import ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--k", type=int, default=1024)
    ap.add_argument("--rows", type=int, default=103227021)
    ap.add_argument("--vocab", type=int, default=267744)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    vals = torch.randint(0, a.vocab, (a.rows,), generator=g, device=dev, dtype=torch.int32)
    ids = torch.randint(0, a.rows, (a.n, a.k), generator=g, device=dev, dtype=torch.int64)
    sims = torch.sort(torch.rand(a.n, a.k, generator=g, device=dev) * 0.7 + 0.2, dim=1, descending=True).values.contiguous()
    tg = vals[ids[:, 3]].long()
    lm = torch.log(torch.rand(a.n, generator=g, device=dev) * 0.9 + 0.01)
    kv = vals[ids].contiguous()
    junk = torch.empty(1 << 28, device=dev, dtype=torch.float32)                     # 1 GiB: flushes L2 / Infinity Cache between runs

    def timeit(f, flush):
        f()
        ts = []
        for _ in range(a.reps):
            if flush:
                junk.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            f()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        return ts[len(ts) // 2], ts[0]

    ref = ops.knn_interp(lm, sims, ids, tg, 0.01, 0.25, vals=vals, vals_tag=False)
    tag = ops.knn_interp(lm, sims, ids, tg, 0.01, 0.25, vals=vals, vals_tag=True, bucketed=False)
    assert all(torch.equal(x, y) for x, y in zip(ref, tag)), "tag path differs"
    rt = ops.knn_interp(lm, sims, ids, tg, 0.01, 0.25, vals=vals, vals_tag=True, bucketed=True)
    assert all(torch.equal(x, y) for x, y in zip(ref, rt)), "routed path differs"
    for flush in (True, False):
        for name, f in (("4-byte label gather", lambda: ops.knn_interp(lm, sims, ids, tg, 0.01, 0.25, vals=vals, vals_tag=False)),
                        ("1-byte tag gather  ", lambda: ops.knn_interp(lm, sims, ids, tg, 0.01, 0.25, vals=vals, vals_tag=True, bucketed=False)),
                        ("routed look-ups    ", lambda: ops.knn_interp(lm, sims, ids, tg, 0.01, 0.25, vals=vals, vals_tag=True, bucketed=True)),
                        ("labels delivered   ", lambda: ops.knn_interp(lm, sims, ids, tg, 0.01, 0.25, knn_vals=kv))):
            med, mn = timeit(f, flush)
            print(f"{name} caches {'flushed' if flush else 'warm   '}: median {med:7.1f} us  min {mn:7.1f} us   "
                  f"({a.n * a.k * 16 / med / 1e3:7.1f} GB/s algorithmic)")


if __name__ == "__main__":
    main()
