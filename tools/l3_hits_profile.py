#!/usr/bin/env python3
"""Per-kernel times of the 3-layer step in the all-hits regime of the centre-state cache (every context group's states cached):
what is left is the tgt side + the cache reads.  `python tools/l3_hits_profile.py [blocks]`; under rocprofv3 --kernel-trace --stats
the same loop gives the per-kernel-name table."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gnnlm_amd import _lib, ops

sys.argv = [sys.argv[0], "--layers", "3", "--blocks", sys.argv[1] if len(sys.argv) > 1 else "16", "--n-store", os.environ.get("N_STORE", "20000000"), "--pool", "2"]
args = bench.parse()
dev = torch.device("cuda:0")
eng, shard, sharded, cpu_model, (d, vocab) = bench.build(args, dev, 0, 1)
batches = bench.make_batches(args, dev, 0, d, vocab)
g = torch.Generator(device=dev); g.manual_seed(5)
pool = torch.randint(0, args.n_store, (200_000,), generator=g, device=dev)
for b in batches:
    b.ids = pool[torch.randint(0, pool.numel(), b.ids.shape, generator=g, device=dev)]
eng.hgt.state_cache_gib = 16.0
acc = torch.zeros(1, device=dev, dtype=torch.float64)
step = lambda b: ops.masked_sum_f64(eng.score(b, 0.25, 0.01)["logp"], None, acc)
for b in batches:
    step(b)                       # cold: fills the cache
for b in batches:
    step(b)
torch.cuda.synchronize()
print("cache stats", eng.hgt.state_cache.stats, "last groups", eng.hgt.last_groups)
n = 10
t0 = time.perf_counter()
for i in range(n):
    step(batches[i % len(batches)])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
tok = args.blocks * args.tokens_per_sample
print(f"all hits: {dt * 1e3:.3f} ms / step, {tok / dt:.0f} tokens/s")
_lib.profile_begin()
for i in range(4):
    step(batches[i % len(batches)])
torch.cuda.synchronize()
k = _lib.profile_end()
tot = sum(v["total_ms"] for v in k.values())
for name, v in sorted(k.items(), key=lambda kv: -kv[1]["total_ms"]):
    print(f"{name:28s} {v['launches'] / 4:6.1f} launches/step  {v['total_ms'] / 4:8.3f} ms/step")
print("sum of kernel times", tot / 4, "ms/step")
