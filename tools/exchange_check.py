#!/usr/bin/env python3
"""Exercise of the sharded-store exchange over RCCL: the routing, the HIP owner-side gather and the fetched-codes path
of the HGT must give the same result as the direct path on a replicated store -- with any number of ranks
(`python tools/exchange_check.py` = one rank; `python -m torch.distributed.run --nproc-per-node N ...` = N ranks),
in the exact (variable-split) and the padded (fixed-capacity, sync-free) mode, slot by slot and through the halo layout."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np, torch, torch.distributed as dist
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
# GNNLM_CHECK_BACKEND=gloo GNNLM_CHECK_DEVICE=0: several ranks on ONE GPU (the collectives staged through the host): the routing,
# the bucketing kernels, the owner-side gather and the consumers with world > 1 on a one-GPU box
backend = os.environ.get("GNNLM_CHECK_BACKEND", "nccl")
if backend != "nccl":
    os.environ["GNNLM_TEST_HOST_STAGED"] = "1"
dev = torch.device("cuda", int(os.environ.get("GNNLM_CHECK_DEVICE", os.environ.get("LOCAL_RANK", "0")))); torch.cuda.set_device(dev)
if backend == "nccl":
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
else:
    dist.init_process_group(backend, rank=rank, world_size=world)
from gnnlm_amd.dist import PeerMappedFetcher, Shard, ShardedFetcher
from gnnlm_amd.hgt import CodeStore
from gnnlm_amd.synthetic import make_problem, build_engine, to_batch
for L in (1, 2):
    prob = make_problem(n_store=5000, d=64, n_heads=4, M=16, dsub=4, vocab=600, cutoff=[100, 300], T=16, kg=8,
                        left=2, right=2, n_layers=L, k=32, seed=10 * L + rank, n_blocks=2)
    eng = build_engine(prob, dev)                       # replicated store: the reference
    b = to_batch(prob["block"], dev)
    ref = eng.score(b, 0.25, 0.01)
    shard = Shard(prob["n_store"], world, rank)
    full = eng.store
    # the table itself is the same on every rank (seeded by L only); the requests differ per rank
    tab = make_problem(n_store=5000, d=64, n_heads=4, M=16, dsub=4, vocab=600, cutoff=[100, 300], T=16, kg=8, left=2, right=2,
                       n_layers=L, k=32, seed=10 * L, n_blocks=2)
    codes = torch.from_numpy(tab["codes"]).to(dev); vals = torch.from_numpy(tab["vals"]).to(dev)
    full.codes, full.vals = codes, vals                 # direct path on the common table
    ref = eng.score(b, 0.25, 0.01)
    # L = 2 needs every slot of a context group: once slot by slot, once through the halo layout (the shard also holds the
    # two rows before / after its range; one request per group, answered by the centre's owner)
    for halo in ((0, 0),) if L == 1 else ((0, 0), (2, 2)):
        hs = Shard(prob["n_store"], world, rank, halo_left=halo[0], halo_right=halo[1])
        sl = slice(hs.store_row0, hs.store_row0 + hs.store_rows)
        part = CodeStore(codes=codes[sl].contiguous(), centroids=full.centroids, n_store=full.n_store, row0=hs.store_row0,
                         vals=vals[sl].contiguous(), A=full.A, b=full.b)
        for mode in ("exact", "padded", "peer"):
            # "peer": no collective at all -- the shards are mapped into every rank (HIP IPC) and gathered directly
            f = PeerMappedFetcher(part, hs, share_vals=True) if mode == "peer" else ShardedFetcher(part, hs, mode=mode)
            b.fetched_codes, b.fetched_valid, b.fetched_index = f.fetch_codes(b.ids, 2, 2, centres_only=(L == 1))
            b.fetched_centres_only = (L == 1)
            b.knn_vals = f.fetch_knn_vals(b.knn_ids)
            out = eng.score(b, 0.25, 0.01)
            torch.cuda.synchronize()
            f.check()
            assert torch.equal(out["logp"], ref["logp"]) and torch.equal(out["recall"], ref["recall"]), (L, mode, halo)
            if rank == 0:
                print("exchange path == direct path, L =", L, mode, "halo" if halo[0] else "slots", "ranks", world)
        # no fetch step at all: the engine's kernels read every code row from its owner's (mapped) shard themselves
        from gnnlm_amd.engine import GnnLmEngine
        f = PeerMappedFetcher(part, hs, share_vals=True)
        eng_m = GnnLmEngine(eng.hgt, eng.asm, f.mapped_store(), eng.left, eng.right)
        b.fetched_codes = b.fetched_valid = b.fetched_index = None
        b.fetched_centres_only = False
        b.knn_vals = f.fetch_knn_vals(b.knn_ids)
        out = eng_m.score(b, 0.25, 0.01)
        torch.cuda.synchronize()
        assert torch.equal(out["logp"], ref["logp"]) and torch.equal(out["recall"], ref["recall"]), (L, "mapped", halo)
        if rank == 0:
            print("mapped shards == direct path, L =", L, "halo" if halo[0] else "slots", "ranks", world)
dist.destroy_process_group()
