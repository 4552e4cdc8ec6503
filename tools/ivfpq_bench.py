#!/usr/bin/env python3
"""IVF-PQ search throughput at the reference's index shape (OPQ64_1024,IVF4096,PQ64, nprobe 32, k = 1024) over a
synthetic 103,227,021-key index (uniform lists, random codes / centroids: shape-true, content-free)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnnlm_amd.synthetic import synthetic_ivfpq_index


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    N = int(os.environ.get("N", 103227021)); n = int(os.environ.get("NQ", 8192)); k = int(os.environ.get("K", 1024))
    idx = synthetic_ivfpq_index(N, 1024, 4096, 64, dev, skew=float(os.environ.get("SKEW", 0.0)), dense_probes=(int(os.environ["DENSE"]) if "DENSE" in os.environ else None), cand_cap=(int(os.environ["CAP"]) if "CAP" in os.environ else None))
    if os.environ.get("PLAIN"):
        idx.packed_codes = None                            # the row-major kernels (A/B)
    torch.manual_seed(0)
    q = torch.randn(n, 1024, device=dev); q = q / q.norm(dim=1, keepdim=True)
    from gnnlm_amd import _lib
    idx.search_device(q, k); torch.cuda.synchronize()      # same shapes as the timed call
    _lib.profile_begin()
    t0 = time.perf_counter(); v, i = idx.search_device(q, k); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    prof = _lib.profile_end()
    st = {k_: (float(v_.item()) if torch.is_tensor(v_) else v_) for k_, v_ in idx.stats.items()}
    print(f"scan={'mfma' if idx.tiles is not None else 'f32'} cand_cap={idx.cand_cap} pairs/query={st['pairs'] / n:.0f} survivors/query={st['survivors'] / n:.0f} "
          f"candidates/query={st['candidates'] / n:.0f} filter groups={st.get('groups', 0):.0f} searched again={st['requeried']}")
    print(f"IVF-PQ search: {n} queries, k={k}, nprobe={idx.nprobe}, N={N}: {dt * 1e3:.1f} ms = {n / dt:.0f} queries/s")
    for kn, e in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
        print(f"   {kn:24s} {e['launches']:4d} launches {e['total_ms']:9.2f} ms")
    reps = int(os.environ.get("REPS", 0))                 # A/B runs: medians over REPS more searches
    if reps:
        import statistics
        per = {}
        tot = []
        for _ in range(reps):
            _lib.profile_begin()
            t0 = time.perf_counter(); idx.search_device(q, k); torch.cuda.synchronize(); tot.append((time.perf_counter() - t0) * 1e3)
            for kn, e in _lib.profile_end().items():
                per.setdefault(kn, []).append(e["total_ms"])
        print(f"medians of {reps}: search {statistics.median(tot):.2f} ms; " + ", ".join(
            f"{kn} {statistics.median(v):.2f}" for kn, v in sorted(per.items(), key=lambda kv: -statistics.median(kv[1]))[:4]))
