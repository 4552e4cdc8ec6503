#!/usr/bin/env python3
"""Single-GPU exercise of the sharded-store exchange over RCCL (world size 1): the routing, the HIP
owner-side gather and the fetched-codes path of the HGT must give the same result as the direct path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np, torch, torch.distributed as dist
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from gnnlm_amd.dist import Shard, ShardedFetcher
from gnnlm_amd.synthetic import make_problem, build_engine, to_batch
for L in (1, 2):
    prob = make_problem(n_store=5000, d=64, n_heads=4, M=16, dsub=4, vocab=600, cutoff=[100, 300], T=16, kg=8,
                        left=2, right=2, n_layers=L, k=32, seed=L, n_blocks=2)
    eng = build_engine(prob, dev)
    b = to_batch(prob["block"], dev)
    ref = eng.score(b, 0.25, 0.01)
    f = ShardedFetcher(eng.store, Shard(prob["n_store"], 1, 0))
    b.fetched_codes, b.fetched_valid, b.fetched_index = f.fetch_codes(b.ids, 2, 2, centres_only=(L == 1))
    b.fetched_centres_only = (L == 1)
    b.knn_vals = f.fetch_knn_vals(b.knn_ids)
    out = eng.score(b, 0.25, 0.01)
    torch.cuda.synchronize()
    assert torch.equal(out["logp"], ref["logp"]) and torch.equal(out["recall"], ref["recall"]), L
    print("exchange path == direct path, L =", L)
dist.destroy_process_group()
