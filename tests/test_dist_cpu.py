"""The sharded-store exchange (gnnlm_amd/dist.py) over gloo with world_size 2 and 3 on the CPU.
The owner-side lookup is injected (numpy indexing of the rank's shard, i.e. the oracle's gather);
on the GPUs the same function runs over RCCL with the HIP gather kernel."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gnnlm_amd.dist import (Shard, bucket_capacity, bucket_padded_torch, exchange_fetch, exchange_fetch_groups,
                            exchange_fetch_padded, slot_rows)
from oracle import graph as og


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_store, M, seed, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rs = np.random.RandomState(seed)
        codes = rs.randint(0, 256, size=(n_store, M)).astype(np.uint8)      # same table on every rank
        vals = rs.randint(0, 1000, size=n_store).astype(np.int32)
        shard = Shard(n_store, world, rank)
        lo, hi = shard.row0, shard.row0 + shard.n_local
        my_codes, my_vals = codes[lo:hi], vals[lo:hi]

        def gather_codes(rows):
            r = rows.numpy()
            ok = (r >= lo) & (r < hi)
            assert ok.all() or ((r[~ok] < 0) | (r[~ok] >= n_store)).all(), "received a row this rank does not own"
            out = np.zeros((len(r), M), np.uint8)
            out[ok] = my_codes[r[ok] - lo]
            return torch.from_numpy(out)

        def gather_vals(rows):
            r = rows.numpy()
            ok = (r >= lo) & (r < hi)
            out = np.full(len(r), -1, np.int32)
            out[ok] = my_vals[r[ok] - lo]
            return torch.from_numpy(out)

        rs2 = np.random.RandomState(100 + rank)                              # different requests per rank
        ids = rs2.randint(0, n_store, size=(7, 5)).astype(np.int64)
        ids[0, 0], ids[1, 1], ids[2, 2], ids[3, :] = -1, 0, n_store - 1, -1
        for left, right in [(0, 0), (2, 2), (1, 0)]:
            rows = slot_rows(torch.from_numpy(ids), left, right, n_store)
            ref_rows, ref_valid = og.slot_layout(ids, n_store, left, right)
            assert np.array_equal(rows.numpy(), ref_rows.reshape(-1))
            got = exchange_fetch(rows, shard, gather_codes).numpy()
            v = ref_valid.reshape(-1)
            assert np.array_equal(got[v], codes[ref_rows.reshape(-1)[v]]) and not got[~v].any()
            payload, index = exchange_fetch(rows, shard, gather_codes, unpermute=False)     # what the HIP consumers use
            assert np.array_equal(payload.numpy()[index.numpy()], got)
        knn = rs2.randint(0, n_store, size=(6, 9)).astype(np.int64)
        knn[0, -2:] = -1
        rows = torch.from_numpy(np.where(knn < 0, knn + n_store, knn).reshape(-1))
        got = exchange_fetch(rows, shard, gather_vals).numpy().reshape(knn.shape)
        assert np.array_equal(got, vals[knn])                                # numpy wrap of -1 included
        # ragged / empty requests
        empty = exchange_fetch(torch.zeros(0, dtype=torch.int64), shard, gather_vals)
        assert empty.numel() == 0
        # fixed-capacity (sync-free) exchange: same answers; rows that are not rows of the store and overflowed requests
        # point at the zero row; an empty shard / a rank that receives only padding must not stall its peers
        for left, right in [(0, 0), (2, 2)]:
            rows = slot_rows(torch.from_numpy(ids), left, right, n_store)
            ref_rows, ref_valid = og.slot_layout(ids, n_store, left, right)
            cap = bucket_capacity(rows.numel(), world)
            back, index, ovf = exchange_fetch_padded(rows, shard, gather_codes, cap)
            assert int(ovf) == 0 and back.shape[0] == world * cap + 1 and not back[-1].any()
            got = back.numpy()[index.numpy()]
            v = ref_valid.reshape(-1)
            assert np.array_equal(got[v], codes[ref_rows.reshape(-1)[v]]) and not got[~v].any()
        # every rank asks the same owner for more than fits: the overflow is counted, the rest is still right
        own0 = torch.arange(0, min(10, shard.per), dtype=torch.int64)
        back, index, ovf = exchange_fetch_padded(own0, shard, gather_codes, 4)
        assert int(ovf) == own0.numel() - 4
        kept = index.numpy() < world * 4
        assert kept.sum() == 4 and np.array_equal(back.numpy()[index.numpy()][kept], codes[own0.numpy()[kept]])
        empty_back, empty_idx, ovf = exchange_fetch_padded(torch.zeros(0, dtype=torch.int64), shard, gather_codes, 64)
        assert empty_idx.numel() == 0 and int(ovf) == 0
        # halo layout: the shard also holds the l / r rows around its range, one request per context group, the centre's
        # owner answers with all of its slots (ids on the links / (1 + l + r))
        for left, right in [(2, 2), (1, 0), (0, 3)]:
            hs = Shard(n_store, world, rank, halo_left=left, halo_right=right)
            hlo, hhi = hs.store_row0, hs.store_row0 + hs.store_rows
            held = codes[hlo:hhi]
            delta = np.array([0] + list(range(-left, 0)) + list(range(1, right + 1)))
            asked = []

            def gather_groups(centres):
                c = centres.numpy()
                asked.append(len(c))
                rows = c[:, None] + delta[None, :]
                ok = (c[:, None] >= 0) & (rows >= 0) & (rows < n_store)
                assert ((c < 0) | ((c >= hs.row0) & (c < hs.row0 + hs.n_local))).all(), "a centre this rank does not own"
                assert ((rows[ok] >= hlo) & (rows[ok] < hhi)).all(), "a slot outside the shard and its halo"
                out = np.zeros(rows.shape + (M,), np.uint8)
                out[ok] = held[rows[ok] - hlo]
                return torch.from_numpy(out)

            ref_rows, ref_valid = og.slot_layout(ids, n_store, left, right)
            v = ref_valid.reshape(-1)
            want = codes[ref_rows.reshape(-1)[v]]
            payload, index, ovf = exchange_fetch_groups(torch.from_numpy(ids), left, right, hs, gather_groups)
            got = payload.numpy()[index.numpy()]
            assert ovf is None and np.array_equal(got[v], want) and not got[~v].any()
            cap = bucket_capacity(ids.size, world)
            payload, index, ovf = exchange_fetch_groups(torch.from_numpy(ids), left, right, hs, gather_groups, cap=cap)
            got = payload.numpy()[index.numpy()]
            assert int(ovf) == 0 and np.array_equal(got[v], want) and not got[~v].any()
            assert payload.shape[0] == (world * cap + 1) * (1 + left + right)
            # merged requests (round 5: equal context groups merged BEFORE the exchange): the requester sends its DISTINCT centre
            # rows -- a 1-D list whose tail beyond the device-side count is -1 (fixed-capacity mode sends the whole list, the -1s
            # are not requests), or exactly the counted prefix (exact mode); a rank whose groups were all cache hits asks nothing
            flat = ids.reshape(-1)
            distinct = np.unique(flat[(flat >= 0) & (flat < n_store)])
            centres = np.full(flat.size, -1, np.int64)
            centres[:len(distinct)] = distinct
            ref_rows, ref_valid = og.slot_layout(distinct.reshape(-1, 1), n_store, left, right)
            want_m = codes[ref_rows.reshape(-1)[ref_valid.reshape(-1)]]
            asked.clear()
            payload, index, _ = exchange_fetch_groups(torch.from_numpy(distinct), left, right, hs, gather_groups)   # exact: the prefix
            got = payload.numpy()[index.numpy()]
            assert np.array_equal(got[ref_valid.reshape(-1)], want_m) and not got[~ref_valid.reshape(-1)].any()
            cap_m = bucket_capacity(len(distinct), world)                          # buckets sized from the distinct count, not from ids.size
            payload, index, ovf = exchange_fetch_groups(torch.from_numpy(centres), left, right, hs, gather_groups, cap=cap_m)
            got = payload.numpy()[index.numpy()].reshape(flat.size, 1 + left + right, M)
            assert int(ovf) == 0 and not got[len(distinct):].any()                 # the -1 tail: zero rows, no request
            got = got[:len(distinct)].reshape(-1, M)
            assert np.array_equal(got[ref_valid.reshape(-1)], want_m) and not got[~ref_valid.reshape(-1)].any()
            if rank == 0:                                                           # rank 0: every group a cache hit this step -- it still serves its peers
                payload, index, _ = exchange_fetch_groups(torch.zeros(0, dtype=torch.int64), left, right, hs, gather_groups)
                assert index.numel() == 0
            else:
                exchange_fetch_groups(torch.from_numpy(distinct[:3]), left, right, hs, gather_groups)
        # final reduction of (score_sum, count) as the eval driver does it
        t = torch.tensor([float(rank + 1), 10.0 * (rank + 1)], dtype=torch.float64)
        dist.all_reduce(t)
        assert t.tolist() == [sum(range(1, world + 1)), 10.0 * sum(range(1, world + 1))]
        q.put((rank, "ok"))
    except Exception as e:                                                   # pragma: no cover
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_store", [(2, 1001), (3, 50)])
def test_exchange_fetch_gloo(world, n_store):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_store, 8, 5, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


def _virtual_rows(rows, M):
    """Content of the (virtual) code table at global rows: a function of the row alone, so that every rank can both serve its
    shard and check what it receives without anybody holding 103 M rows."""
    r = np.asarray(rows, dtype=np.int64)
    return (((r[..., None] * 2654435761 + np.arange(M, dtype=np.int64) * 40503) >> 7) & 255).astype(np.uint8)


def _worker_full_size(rank, world, port, q):
    """BASELINE.json configs[2] at its REAL shard arithmetic on 8 ranks: N = 103,227,021 keys in 8 key ranges with a halo of
    l = r = 2 rows (`Shard(..., halo_left=2, halo_right=2)`: per = 12,903,378, the last range 3 rows short), k_g = 1024 graph
    neighbours per token, one request per context group answered by the centre's owner -- exact and fixed-capacity exchange,
    uniform ids plus every row within 3 of a range boundary, the store's two ends and -1.  The table is virtual (`_virtual_rows`)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        N, M, left, right, kg, T = 103_227_021, 8, 2, 2, 1024, 8
        hs = Shard(N, world, rank, halo_left=left, halo_right=right)
        assert hs.per == 12_903_378 and hs.row0 == rank * hs.per
        assert hs.n_local == (hs.per if rank < world - 1 else N - (world - 1) * hs.per) and (rank < world - 1 or hs.n_local == hs.per - 3)
        assert hs.store_row0 == max(0, hs.row0 - left) and hs.store_row0 + hs.store_rows == min(N, hs.row0 + hs.n_local + right)
        hlo, hhi = hs.store_row0, hs.store_row0 + hs.store_rows
        delta = np.array([0] + list(range(-left, 0)) + list(range(1, right + 1)))
        served = []

        def gather_groups(centres):
            c = centres.numpy()
            served.append(len(c))
            rows = c[:, None] + delta[None, :]
            ok = (c[:, None] >= 0) & (rows >= 0) & (rows < N)
            assert ((c < 0) | ((c >= hs.row0) & (c < hs.row0 + hs.n_local))).all(), "a centre this rank does not own"
            assert ((rows[ok] >= hlo) & (rows[ok] < hhi)).all(), "a slot outside the shard and its halo"
            out = np.zeros(rows.shape + (M,), np.uint8)
            out[ok] = _virtual_rows(rows[ok], M)
            return torch.from_numpy(out)

        rs = np.random.RandomState(1000 + rank)
        ids = rs.randint(0, N, size=(T, kg)).astype(np.int64)
        edges = np.concatenate([np.arange(b - 3, b + 4) for b in range(hs.per, N, hs.per)] + [np.arange(0, 4), np.arange(N - 4, N), [-1, -1]])    # (ids are rows of the train datastore or -1: nothing beyond N - 1)
        ids.reshape(-1)[:len(edges)] = edges                                # every boundary of every range, from every rank
        own = hs.owner(torch.from_numpy(ids.reshape(-1))).numpy()
        inr = (ids.reshape(-1) >= 0) & (ids.reshape(-1) < N)
        assert np.array_equal(own[inr], np.minimum(ids.reshape(-1)[inr] // hs.per, world - 1)) and (own[~inr] == rank).all()
        ref_rows, ref_valid = og.slot_layout(ids, N, left, right)
        v = ref_valid.reshape(-1)
        want = _virtual_rows(ref_rows.reshape(-1)[v], M)
        payload, index, ovf = exchange_fetch_groups(torch.from_numpy(ids), left, right, hs, gather_groups)              # exact splits
        got = payload.numpy()[index.numpy()]
        assert ovf is None and np.array_equal(got[v], want) and not got[~v].any()
        cap = bucket_capacity(ids.size, world)                                                                           # fixed capacity: 2 equal-split all-to-alls
        assert cap % 64 == 0 and cap >= 1.25 * ids.size / world
        payload, index, ovf = exchange_fetch_groups(torch.from_numpy(ids), left, right, hs, gather_groups, cap=cap)
        got = payload.numpy()[index.numpy()]
        assert int(ovf) == 0 and payload.shape[0] == (world * cap + 1) * (1 + left + right)
        assert np.array_equal(got[v], want) and not got[~v].any()
        # what the links carry per rank and step at this shape: 8-B requests out, (1 + l + r) M-byte groups back, 7/8 remote
        t = torch.tensor([float(sum(served))], dtype=torch.float64)
        dist.all_reduce(t)
        assert t.item() >= 2 * world * inr.sum() * 0.99                    # every in-range group was served once per exchange, somewhere
        q.put((rank, "ok"))
    except Exception as e:                                                   # pragma: no cover
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_exchange_world8_at_the_full_store_shard_arithmetic():
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_full_size, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


def test_bucket_padded_layout():
    s = Shard(1000, 3, 1)
    rows = torch.tensor([5, 999, -1, 400, 1000, 333, 334, 700, 2], dtype=torch.int64)
    send, index, ovf = bucket_padded_torch(rows, s, 2)
    assert send.tolist() == [5, 333, 400, 334, 999, 700] and ovf.tolist() == [1]          # owner 0 got 3 requests, 2 fit
    assert index.tolist() == [0, 4, 6, 2, 6, 1, 3, 5, 6]                                  # 6 = the zero row
    assert bucket_capacity(1_048_576, 8) % 64 == 0 and bucket_capacity(1_048_576, 8) >= 1.25 * 131072


def test_shard_geometry():
    for n, w in [(10, 3), (103227021, 8), (5, 8)]:
        tot = 0
        for r in range(w):
            s = Shard(n, w, r)
            assert s.row0 == min(r * s.per, n)
            tot += s.n_local
        assert tot == n
    s = Shard(100, 4, 1)
    rows = torch.tensor([-1, 0, 24, 25, 49, 50, 99, 100, 1000])
    assert s.owner(rows).tolist() == [1, 0, 0, 1, 1, 2, 3, 1, 1]


def test_peer_map_watchdog_kills_a_stuck_rank_and_spares_a_finished_one():
    """`--exchange peer` must fail fast, never hang: the watchdog around the shard mapping is a helper child process that kills the
    rank (non-zero exit, a message on stderr) when the mapping does not return in time, and goes away when it does."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stuck = ("import sys, time; sys.path.insert(0, %r); from gnnlm_amd.dist import _MapWatchdog\n"
             "with _MapWatchdog('mapping (test)', seconds=1.0):\n    time.sleep(60)\nprint('not reached')" % root)
    t0 = time.time()
    p = subprocess.run([sys.executable, "-c", stuck], capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and time.time() - t0 < 30 and "not reached" not in p.stdout
    assert "did not finish within 1 s" in p.stderr and "--exchange padded|exact" in p.stderr
    fine = ("import sys, time; sys.path.insert(0, %r); from gnnlm_amd.dist import _MapWatchdog\n"
            "with _MapWatchdog('mapping (test)', seconds=1.0) as w:\n    pass\ntime.sleep(2.0); assert w.proc.poll() is not None; print('alive')" % root)
    p = subprocess.run([sys.executable, "-c", fine], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "alive" in p.stdout and p.stderr.strip() == ""
