"""The routed label look-ups of the ids-only kNN interpolation (csrc/knn_bucket.hip) at the bench's shape, per call, by the library's own event
profiler; `GNNLM_LIB=<variant>.so` times an A/B build (tools/build_variant.sh kbN knn_bucket.hip -DGNNLM_KB_EXP=N / -DGNNLM_KB_TILE=T)."""
import sys, os, csv, glob
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from gnnlm_amd import ops, _lib
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
n, k, N, V = 8192, 1024, 103227021, 267744
vals = torch.randint(0, V, (N,), generator=g, device=dev, dtype=torch.int32)
ids = torch.randint(0, N, (n, k), generator=g, device=dev, dtype=torch.int64)
sims = torch.rand(n, k, generator=g, device=dev)
tg = vals[ids[:, 3]].long(); lm = torch.zeros(n, device=dev)
f = lambda: ops.knn_interp(lm, sims, ids, tg, 0.01, 0.25, vals=vals, vals_tag=True, bucketed=True)
for _ in range(3): f()
torch.cuda.synchronize()
_lib.profile_begin()
for _ in range(10): f()
torch.cuda.synchronize()
pr = _lib.profile_end()
print(os.environ.get("GNNLM_LIB", "shipped"), {k_: round(v["total_ms"] * 100, 1) for k_, v in pr.items()}, "us per call")
