#!/usr/bin/env python3
"""How many keys would the filter of the IVF-PQ search let through under the formulations VERDICT r03 asked about -- measured, not
argued: for queries against a clustered index of the reference's shape (OPQ64_1024,IVF4096,PQ64, nprobe 32, k = 1024), the number of
(query, key) pairs whose UPPER bound exceeds the query's threshold tau (its exact k-th best score: the most favourable threshold any
threshold pass could deliver), for

  u8      the shipped byte tables: one-sided bound, one delta per query                            (64 B of LDS look-ups per pair)
  u4      4-bit tables, uniform grid per query (16 queries per 8-byte look-up)                     (32 B)
  u4sub   4-bit tables, one grid PER SUB-QUANTIZER (the tightest uniform 4-bit grid)               (32 B)
  u4nu    4-bit tables on a NON-UNIFORM grid: per sub-quantizer the 16 quantiles of its 256 entries, rounded up (one-sided)   (32 B)
  half    two-stage: exact sums of the first 32 sub-quantizers + the best case of the other 32 (their per-table maxima): the
          fraction of pairs that pass stage 1 and need the second half of the look-ups             (32 B + 32 B for the survivors)

    python tools/filter_formulations.py [--keys 2000000] [--queries 64]
Survivors are counted against `need` = the keys that truly score above tau (k per query)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--keys", type=int, default=2_000_000)
    ap.add_argument("--queries", type=int, default=64)
    ap.add_argument("--k", type=int, default=1024)
    a = ap.parse_args()
    from gnnlm_amd.ivfpq import IVFPQIndex
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    N, d, nc = a.keys, 1024, 3000
    centres = torch.randn(nc, d, generator=g, device=dev)
    p = 1.0 / torch.arange(1, nc + 1, device=dev, dtype=torch.float64) ** 0.7
    which = torch.multinomial(p / p.sum(), N, replacement=True, generator=g)
    keys = torch.empty(N, d, device=dev, dtype=torch.float16)
    for s in range(0, N, 1 << 19):
        w = which[s:s + (1 << 19)]
        keys[s:s + (1 << 19)] = (centres[w] + 0.8 * torch.randn(w.shape[0], d, generator=g, device=dev)).to(torch.float16)
    nlist = max(64, min(4096, N // 2000))
    index = IVFPQIndex.build(keys, nlist, 64, device=dev, cosine=True, nprobe=32, iters=6, seed=5)
    q = centres[torch.randint(0, nc, (a.queries,), generator=g, device=dev)] + 0.8 * torch.randn(a.queries, d, generator=g, device=dev)
    q = q / q.norm(dim=1, keepdim=True)
    qr = q @ index.R.t()
    cs = qr @ index.coarse.t()
    probes = cs.topk(32, dim=1).indices
    lut = torch.einsum("nmd,mcd->nmc", qr.view(a.queries, 64, -1), index.pq)                   # [n, 64, 256]
    tot = {k_: 0 for k_ in ("pairs", "need", "u8", "u4", "u4sub", "u4nu", "half")}
    for r in range(a.queries):
        L = lut[r]                                                                              # [64, 256]
        rows = torch.cat([torch.arange(int(index.list_off[l]), int(index.list_off[l + 1]), device=dev) for l in probes[r].tolist()])
        bias = torch.cat([cs[r, l].expand(int(index.list_off[l + 1] - index.list_off[l])) for l in probes[r].tolist()])
        codes = index.list_codes[rows].long()                                                   # [P, 64]
        ar = torch.arange(64, device=dev)[None, :]
        score = bias + L[ar, codes].sum(1)
        if score.numel() < a.k:
            continue
        tau = score.topk(a.k).values[-1]
        lo = L.min(1, keepdim=True).values
        rng = (L.max(1, keepdim=True).values - lo)

        def ub_uniform(levels, per_sub):
            delta = (rng / levels) if per_sub else (rng.max() / levels).expand_as(rng)
            u = torch.clamp(torch.floor((L - lo) / delta.clamp_min(1e-30)), max=levels)
            return bias + (lo + (u + 1) * delta)[ar, codes].sum(1)                              # one-sided: L < lo + (u + 1) delta

        ub8 = ub_uniform(255, False)
        ub4 = ub_uniform(15, False)
        ub4s = ub_uniform(15, True)
        # non-uniform 4-bit grid: level j of sub-quantizer m = the ((j + 1) / 16)-quantile of its 256 entries; an entry is
        # rounded UP to the next level (one-sided), the top level is the maximum
        srt = L.sort(1).values
        levels = srt[:, torch.arange(15, 256, 16, device=dev)]                                  # [64, 16] upper edges
        idx = torch.searchsorted(levels.contiguous(), L.contiguous()).clamp(max=15)
        Lup = torch.gather(levels, 1, idx)
        ub4n = bias + Lup[ar, codes].sum(1)
        first = bias + L[ar[:, :32], codes[:, :32]].sum(1) + L[32:].max(1).values.sum()        # stage 1 of the two-stage filter
        tot["pairs"] += score.numel()
        tot["need"] += int((score >= tau).sum())
        for nm, ub in (("u8", ub8), ("u4", ub4), ("u4sub", ub4s), ("u4nu", ub4n), ("half", first)):
            tot[nm] += int((ub > tau).sum())
    n = a.queries
    print(f"{N} keys, {nlist} lists, {n} queries, nprobe 32, k = {a.k}: pairs/query {tot['pairs'] / n:.0f}, keys above tau {tot['need'] / n:.0f}")
    for nm in ("u8", "u4", "u4sub", "u4nu", "half"):
        print(f"  {nm:6s} survivors/query {tot[nm] / n:10.0f}   = {tot[nm] / max(tot['need'], 1):7.1f} x the keys above tau   = {100.0 * tot[nm] / tot['pairs']:6.2f} % of the pairs")


if __name__ == "__main__":
    main()
