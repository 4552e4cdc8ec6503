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
# ---- multi-layer models: equal context groups are merged ON THE DEVICE before the exchange (one request per distinct centre row;
# with the cross-batch cache only for the rows it lacks) -- bit-identical to the un-merged direct path, fewer bytes on the links
from gnnlm_amd.engine import GnnLmEngine
from gnnlm_amd.hgt import HGT
for L in (2, 3):
    prob = make_problem(n_store=5000, d=64, n_heads=4, M=16, dsub=4, vocab=600, cutoff=[100, 300], T=16, kg=8,
                        left=2, right=2, n_layers=L, k=32, seed=30 + L, n_blocks=2)
    eng = build_engine(prob, dev)
    eng.hgt.dedup_groups = False                        # the reference of this section: the un-merged graph on the whole table
    per = -(-prob["n_store"] // world)
    rs = np.random.RandomState(100 * L + rank)
    # few distinct rows (heavy overlap between tokens, blocks and batches), some of them at the shard boundaries and store ends
    pool = np.concatenate([rs.randint(0, prob["n_store"], 40), [0, 1, prob["n_store"] - 1, per - 1, per, per + 1, min(prob["n_store"] - 1, 2 * per)]])
    batches = []
    for j in range(3):
        bb = to_batch(prob["block"], dev)
        ids = pool[rs.randint(0, len(pool) - 10 * (j == 0), size=tuple(bb.ids.shape))].astype(np.int64)     # (batch 0 never sees the last rows: later misses)
        ids[rs.rand(*ids.shape) < 0.05] = -1
        bb.ids = torch.from_numpy(ids).to(dev)
        batches.append(bb)
    refs = [eng.score(bb, 0.25, 0.01)["logp"].clone() for bb in batches]
    full = eng.store
    for halo in ((2, 2), (0, 0)):
        hs = Shard(prob["n_store"], world, rank, halo_left=halo[0], halo_right=halo[1])
        sl = slice(hs.store_row0, hs.store_row0 + hs.store_rows)
        part = CodeStore(codes=full.codes[sl].contiguous(), centroids=full.centroids, n_store=full.n_store, row0=hs.store_row0,
                         vals=full.vals, A=full.A, b=full.b)
        part.vals_row0 = 0                              # (labels replicated)
        for mode in ("exact", "padded", "peer", "mapped"):
            if mode == "padded" and not halo[0]:
                continue                                # (the fixed-capacity exchange of merged groups needs halo shards)
            for cache_slots in (None, 4096):
                hgt = HGT(in_dim=64, hidden_dim=64, out_dim=64, n_layers=L, n_heads=4)
                hgt.load_state_dict(eng.hgt.state_dict())
                assert hgt.dedup_groups
                hgt.state_cache_gib, hgt.state_cache_slots = (0.0, None) if cache_slots is None else (1.0, cache_slots)
                if mode == "mapped":
                    f = None
                    e2 = GnnLmEngine(hgt, eng.asm, PeerMappedFetcher(part, hs).mapped_store(), eng.left, eng.right)
                    e2.store.vals_row0 = 0
                else:
                    f = PeerMappedFetcher(part, hs) if mode == "peer" else ShardedFetcher(part, hs, mode=mode)
                    e2 = GnnLmEngine(hgt, eng.asm, part, eng.left, eng.right, fetcher=f)
                computed = []
                for bb, ref in zip(batches + batches[:1], refs + refs[:1]):
                    out = e2.score(bb, 0.25, 0.01)
                    torch.cuda.synchronize()
                    assert torch.equal(out["logp"], ref), (L, mode, halo, cache_slots)
                    computed.append(hgt.last_groups)
                if f is not None:
                    f.check()
                n_all = computed[0][0]
                assert all(c[1] < n_all // 4 for c in computed), computed             # requests merged: far fewer groups than neighbours
                if cache_slots is not None:
                    assert computed[-1][1] == 0 and computed[1][1] < computed[0][1]   # batch 0 again: every group cached
                if mode == "exact":
                    per_request = (8 + 5 * 16) if halo[0] else 5 * (8 + 16)       # one request per group (halo layout) / per slot
                    unmerged = sum(int(c[0] * (world - 1) / world) * 2 * per_request for c in computed)
                    assert world == 1 or f.link_bytes < unmerged // 4, (f.link_bytes, unmerged)
                if rank == 0:
                    print("merged exchange == direct path, L =", L, mode, "halo" if halo[0] else "slots", "cache" if cache_slots else "no cache",
                          "ranks", world, "groups computed", [c[1] for c in computed], "of", n_all)
dist.destroy_process_group()
