"""Range-sharded datastore over the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference has no multi-GPU eval path; this is the design BASELINE.json's
north_star asks for.  Token blocks are data parallel (independent, no collective); the only exchange
is the row fetch from the range-sharded PQ code table / label table:

    rank g owns rows [g*per, (g+1)*per)   with per = ceil(n_store / world)

Per step (exact mode): bucket the requested rows by owner (HIP histogram + scatter, no sort) -> all_to_all
of the counts -> all_to_all of the row ids (8 B/row) -> every owner gathers its rows with the HIP gather
kernel -> all_to_all of the payload back (M = 128 B/row of codes).  Padded mode (`exchange_fetch_padded`)
ships fixed-capacity buckets instead: two equal-split all-to-alls, no counts, no host synchronisation.  The payload stays in bucketed order:
the consumer kernels read row ``index[s]`` for request s, so the 128-B rows are not shuffled again.
The label table (4 B/row, 413 MB for WikiText-103) is replicated by default -- sharding it would add a
k = 1024-per-token exchange for 3 % of the store's bytes (``bench.py --shard-vals`` does it anyway).  xGMI is point-to-point
(7 links x ~153 GB/s per GPU): an all-to-all drives all 7 links at once, unlike a ring, so one fused
exchange per table per step is the right shape.  Row validity needs no communication: a row is valid
iff 0 <= row < n_store, which the requester knows.

Halo layout: with L > 1 the consumers need every slot of a context group (centre, l rows
before, r rows after; token_block_dataset.py:358,380-385).  A shard built with ``Shard(..., halo_left=l,
halo_right=r)`` also keeps the l rows before and the r rows after its range, so the OWNER OF THE CENTRE can answer
the whole group: one 8-B request per group instead of 1 + l + r, answered with (1 + l + r) * M bytes
(`exchange_fetch_groups`; ids on the links / (1 + l + r), bucketing work too, payload unchanged).

The function is backend-agnostic (RCCL on the GPUs; the CPU tests run it over gloo with a numpy
gather injected for the owner-side lookup).
"""
from dataclasses import dataclass
from typing import Callable

import torch
import torch.distributed as dist


@dataclass
class Shard:
    n_store: int
    world: int
    rank: int
    halo_left: int = 0          # rows kept in front of / behind the owned range (copies of the neighbours' edge rows)
    halo_right: int = 0

    @property
    def per(self):
        return -(-self.n_store // self.world)

    @property
    def row0(self):
        return min(self.rank * self.per, self.n_store)

    @property
    def n_local(self):
        return max(0, min((self.rank + 1) * self.per, self.n_store) - self.row0)

    @property
    def store_row0(self):
        """First row this rank HOLDS (owned range plus halo)."""
        return max(0, self.row0 - self.halo_left) if self.n_local else self.row0

    @property
    def store_rows(self):
        if not self.n_local:
            return 0
        return min(self.n_store, self.row0 + self.n_local + self.halo_right) - self.store_row0

    def owner(self, rows: torch.Tensor) -> torch.Tensor:
        """Owning rank of each global row; out-of-range rows (incl. -1) are kept local."""
        ok = (rows >= 0) & (rows < self.n_store)
        own = torch.div(rows.clamp(min=0), self.per, rounding_mode="floor").clamp(max=self.world - 1)
        return torch.where(ok, own, torch.full_like(own, self.rank))


def _all_to_all(out: torch.Tensor, inp: torch.Tensor, out_splits=None, in_splits=None, group=None):
    """all_to_all_single; a process group that cannot move device memory (gloo: used to run TWO ranks on ONE GPU in the
    tests -- RCCL refuses two ranks per device) is served through pinned host copies of the same buffers."""
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        import os
        if os.environ.get("GNNLM_TEST_HOST_STAGED") != "1":          # never silently: the product's transport is RCCL
            raise RuntimeError("sharded exchange: device tensors over a gloo process group (host-staged copies) are a test "
                               "transport only -- use backend 'nccl' (RCCL), or set GNNLM_TEST_HOST_STAGED=1 in a test")
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)


def bucket_torch(rows: torch.Tensor, shard: Shard):
    """Backend-agnostic bucketing (stable sort by owner).  -> (counts [world] i64, send_rows, inv)."""
    owner = shard.owner(rows)
    order = torch.sort(owner, stable=True).indices
    inv = torch.empty_like(order)
    inv[order] = torch.arange(order.numel(), device=order.device)
    return torch.bincount(owner, minlength=shard.world).to(torch.int64), rows[order].contiguous(), inv


def bucket_hip(rows: torch.Tensor, shard: Shard):
    """Bucketing by the HIP kernel pair of gnnlm_bucket_rows (histogram + scatter; no sort)."""
    from . import _lib
    n = rows.numel()
    dev = rows.device
    counts = torch.zeros(shard.world, dtype=torch.int64, device=dev)
    cursor = torch.empty(shard.world, dtype=torch.int64, device=dev)
    send_rows = torch.empty(n, dtype=torch.int64, device=dev)
    inv = torch.empty(n, dtype=torch.int32, device=dev)
    _lib.call("gnnlm_bucket_rows", _lib.ptr(rows), n, shard.n_store, shard.per, shard.world, shard.rank,
              _lib.ptr(counts), _lib.ptr(cursor), _lib.ptr(send_rows), _lib.ptr(inv), _lib.stream())
    return counts, send_rows, inv


def exchange_fetch(rows: torch.Tensor, shard: Shard, local_gather: Callable[[torch.Tensor], torch.Tensor],
                   group=None, bucket=bucket_torch, unpermute: bool = True):
    """Fetch ``payload[row]`` for every global row in ``rows`` (int64 [S]) from its owning rank.

    ``local_gather(global_rows) -> [n, C]`` is evaluated on the owner for the rows it receives (all of
    them inside its shard, or out of range, for which it must return zeros).  Returns [S, C] in the
    order of ``rows``; with ``unpermute=False`` returns (payload in bucketed order, index) where
    ``payload[index[s]]`` answers request s -- the consumer kernels apply the index themselves and the
    128-B rows are never shuffled a second time.  Two host syncs per call (the variable split sizes)."""
    S = rows.numel()
    rows = rows.reshape(-1).contiguous()
    counts, send_rows, inv = bucket(rows, shard)
    recv_counts = torch.empty_like(counts)
    _all_to_all(recv_counts, counts, group=group)
    in_splits, out_splits = counts.tolist(), recv_counts.tolist()
    recv_rows = torch.empty(sum(out_splits), dtype=rows.dtype, device=rows.device)
    _all_to_all(recv_rows, send_rows, out_splits, in_splits, group)
    payload = local_gather(recv_rows).contiguous()
    assert payload.shape[0] == recv_rows.numel()
    back = torch.empty((S,) + tuple(payload.shape[1:]), dtype=payload.dtype, device=payload.device)
    _all_to_all(back, payload, in_splits, out_splits, group)
    if not unpermute:
        return back, inv
    return back[inv.long()]


def bucket_capacity(n_requests: int, world: int, slack: float = 1.25) -> int:
    """Slots per owner of the fixed-capacity exchange: the uniform-id expectation n / world plus `slack` and a
    constant (a binomial count exceeds 1.25 n/world + 256 with negligible probability from a few thousand requests
    on), rounded to 64."""
    cap = int(-(-n_requests // world) * slack) + 256
    return (cap + 63) // 64 * 64


def bucket_padded_torch(rows: torch.Tensor, shard: Shard, cap: int):
    """Backend-agnostic fixed-capacity bucketing, no host round trip.
    -> (send [world*cap] i64 with -1 in unused slots, index [S] i32 = slot of request s or world*cap for a request
    that is not sent (not a row of the store) or did not fit, overflow [1] i64 = requests that did not fit)."""
    W = shard.world
    ok_row = (rows >= 0) & (rows < shard.n_store)
    owner = torch.where(ok_row, shard.owner(rows), torch.full_like(rows, W))          # W: "not sent"
    order = torch.sort(owner, stable=True).indices
    counts = torch.bincount(owner, minlength=W + 1)
    start = torch.cumsum(counts, 0) - counts
    pos = torch.empty_like(rows)
    pos[order] = torch.arange(rows.numel(), device=rows.device) - start[owner[order]]
    fits = ok_row & (pos < cap)
    index = torch.where(fits, owner * cap + pos, torch.full_like(rows, W * cap))
    send = torch.full((W * cap + 1,), -1, dtype=rows.dtype, device=rows.device)
    send[index] = torch.where(fits, rows, torch.full_like(rows, -1))
    return send[:W * cap].contiguous(), index.to(torch.int32), (ok_row & ~fits).sum().reshape(1)


def bucket_padded_hip(rows: torch.Tensor, shard: Shard, cap: int):
    """The same by the HIP kernel gnnlm_bucket_rows_padded (one scatter pass)."""
    from . import _lib
    n, dev, W = rows.numel(), rows.device, shard.world
    cursor = torch.empty(W, dtype=torch.int64, device=dev)
    send = torch.empty(W * cap, dtype=torch.int64, device=dev)
    index = torch.empty(n, dtype=torch.int32, device=dev)
    overflow = torch.zeros(1, dtype=torch.int64, device=dev)
    _lib.call("gnnlm_bucket_rows_padded", _lib.ptr(rows), n, shard.n_store, shard.per, W, cap, _lib.ptr(cursor),
              _lib.ptr(send), _lib.ptr(index), _lib.ptr(overflow), _lib.stream())
    return send, index, overflow


def exchange_fetch_padded(rows: torch.Tensor, shard: Shard, local_gather: Callable[[torch.Tensor], torch.Tensor], cap: int,
                          group=None, bucket=bucket_padded_torch):
    """Sync-free exchange: TWO equal-split all-to-alls (row ids out, payload back), no counts exchange and no host
    round trip -- the bucket sizes never reach the host.  Every rank sends `cap` slots to every rank (-1 padded;
    `bucket_capacity` sizes cap from the uniform-id bound, so the padding costs ~25 % more bytes than the exact
    variable-split exchange above and saves its two stream synchronisations).
    -> (payload [world*cap + 1, C] whose last row is zeros, index [S] i32, overflow [1] i64): payload[index[s]]
    answers request s; requests that are not rows of the store, or overflowed their bucket, point at the zero row.
    A non-zero `overflow` means some rows were NOT fetched: the caller must check it (ShardedFetcher.check)."""
    rows = rows.reshape(-1).contiguous()
    send, index, overflow = bucket(rows, shard, cap)
    recv = torch.empty_like(send)
    _all_to_all(recv, send, group=group)
    payload = local_gather(recv).contiguous()                       # [world*cap, ...]; -1 ids give zero rows
    back = torch.zeros((shard.world * cap + 1,) + tuple(payload.shape[1:]), dtype=payload.dtype, device=payload.device)
    _all_to_all(back[:shard.world * cap], payload, group=group)
    return back, index, overflow


def exchange_fetch_groups(ids: torch.Tensor, left: int, right: int, shard: Shard,
                          group_gather: Callable[[torch.Tensor], torch.Tensor], group=None, cap: int = 0,
                          bucket=None):
    """Halo-layout fetch of whole context groups: ONE request (the centre id, 8 B) per group, answered by the centre's
    owner with the group's 1 + left + right rows.  ``group_gather(centres [n]) -> [n, 1 + left + right, C]`` runs on the
    owner (slot order: centre, o - left .. o - 1, o + 1 .. o + right; zero rows for slots outside the store and for
    centres < 0); the owner's shard must hold `left` / `right` halo rows.
    cap = 0: exact variable-split exchange; cap > 0: fixed-capacity sync-free exchange.
    -> (payload [P * n_g, C], index int32 [G * n_g], overflow [1] or None): slot s lives in payload row index[s]."""
    assert shard.halo_left >= left and shard.halo_right >= right, "the shard was built without the halo rows this needs"
    n_g = 1 + left + right
    centres = ids.reshape(-1)
    centres = torch.where((centres >= 0) & (centres < shard.n_store), centres, torch.full_like(centres, -1)).contiguous()
    k = torch.arange(n_g, device=centres.device, dtype=torch.int64)
    if cap:
        back, index, ovf = exchange_fetch_padded(centres, shard, group_gather, cap, group,
                                                 bucket=bucket or bucket_padded_torch)
    else:
        back, index = exchange_fetch(centres, shard, group_gather, group, bucket=bucket or bucket_torch, unpermute=False)
        ovf = None
    assert back.shape[1] == n_g
    slot_index = (index.long().reshape(-1, 1) * n_g + k).reshape(-1).to(torch.int32)
    return back.reshape((back.shape[0] * n_g,) + tuple(back.shape[2:])), slot_index, ovf


def slot_rows(ids: torch.Tensor, left: int, right: int, n_store: int) -> torch.Tensor:
    """Global row of every slot of every group (centre, o-left..o-1, o+1..o+right); -1 if invalid
    (token_block_dataset.py:358,380-385)."""
    delta = torch.tensor([0] + list(range(-left, 0)) + list(range(1, right + 1)), dtype=torch.int64, device=ids.device)
    rows = ids.reshape(-1, 1) + delta
    ok = (ids.reshape(-1, 1) >= 0) & (rows >= 0) & (rows < n_store)
    return torch.where(ok, rows, torch.full_like(rows, -1)).reshape(-1)


class ShardedFetcher:
    """Product-side exchange: the owner-side lookup is the HIP gather kernel."""

    def __init__(self, store, shard: Shard, group=None, mode: str = "exact", slack: float = 1.25):
        """mode "exact": variable-split exchange (two host syncs per call, never drops a row);
        mode "padded": fixed-capacity, sync-free exchange; call :meth:`check` (it synchronises) before trusting a run."""
        from . import ops
        self.ops, self.store, self.shard, self.group = ops, store, shard, group
        assert mode in ("exact", "padded")
        self.mode, self.slack = mode, slack
        self.overflow = None                                # device counter of rows that did not fit (padded mode)
        self.link_bytes = 0                                 # bytes this rank put on / took off its xGMI links so far
        assert store.row0 == shard.store_row0 and store.codes.shape[0] == shard.store_rows
        # fixed-capacity mode: the all-to-alls are equal-split, so EVERY rank must derive the same bucket capacity.  Ranks whose
        # batches differ in size (a ragged last batch, an idle rank) set `fixed_requests` to the agreed largest request count of
        # a step; `group_requests` (calibrate_groups) does the same for merged requests, whose count is data dependent
        self.fixed_requests = None
        self.group_requests = None

    def check(self):
        """Raise if the fixed-capacity exchange dropped requests (skewed ids): rerun with mode='exact' or more slack."""
        if self.overflow is not None and int(self.overflow.item()) > 0:
            raise RuntimeError(f"sharded exchange: {int(self.overflow.item())} row requests did not fit their fixed-capacity "
                               f"buckets (slack {self.slack}); use mode='exact' or a larger slack")

    def _padded(self, rows, gather, row_bytes, requests=None):
        W = self.shard.world
        cap = bucket_capacity(requests or rows.numel(), W, self.slack)
        back, index, ovf = exchange_fetch_padded(rows, self.shard, gather, cap, self.group, bucket=bucket_padded_hip)
        self.overflow = ovf if self.overflow is None else self.overflow + ovf
        self.link_bytes += cap * (W - 1) * 2 * (8 + row_bytes)       # ids out + payload in, and the same served to the peers
        return back, index, cap

    def _account_exact(self, n_rows, row_bytes):
        W = self.shard.world
        self.link_bytes += int(n_rows * (W - 1) / W) * 2 * (8 + row_bytes)     # uniform-id expectation

    def _fetch_groups(self, ids, left, right, cap_requests=None):
        """Halo layout: one request per context group (see exchange_fetch_groups)."""
        st, W, n_g = self.store, self.shard.world, 1 + left + right
        M = st.codes.shape[1]

        def gather(centres):
            return self.ops.pq_gather_decode(st.codes, st.centroids, centres, left, right, n_store=st.n_store, row0=st.row0,
                                             want_x=False, want_codes=True, want_valid=False)["codes"].view(-1, n_g, M)
        G = ids.numel()
        valid = slot_rows(ids, left, right, st.n_store) >= 0
        if self.mode == "padded":
            cap = bucket_capacity((self.fixed_requests or G) if cap_requests is None else cap_requests, W, self.slack)
            codes, index, ovf = exchange_fetch_groups(ids, left, right, self.shard, gather, self.group, cap, bucket_padded_hip)
            self.overflow = ovf if self.overflow is None else self.overflow + ovf
            self.link_bytes += cap * (W - 1) * 2 * (8 + n_g * M)
            valid = valid & (index.long() < W * cap * n_g)
        else:
            codes, index, _ = exchange_fetch_groups(ids, left, right, self.shard, gather, self.group, 0, bucket_hip)
            self._account_exact(G, n_g * M)
        return codes, valid.to(torch.uint8), index

    def calibrate_groups(self, counters, slack=1.15):
        """Fixed-capacity mode, merged requests: size the buckets from a MEASURED distinct-group count instead of the worst case
        (every neighbour its own group).  ``counters``: the device counters of a representative merged step
        (``HGT._last_groups[1]``).  Synchronises and all-reduces (MAX) -- every rank must end up with the same capacity, the
        all-to-alls are equal-split -- so call it in a warm-up phase, never inside a timed region.  A later batch with more
        distinct groups than measured x slack overflows its buckets: counted, and raised by :meth:`check`."""
        t = counters[:1].to(torch.int64).clone()
        if self.shard.world > 1:
            if t.is_cuda and dist.get_backend(self.group) == "gloo":
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.group)
                t = h
            else:
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        self.group_requests = int(int(t.item()) * slack) + 1024
        return self.group_requests

    def fetch_groups(self, centres, left, right, counters):
        """The slots of MERGED context groups (HGT.forward, multi-layer models): ``centres`` int64 [n] are the distinct centre rows
        of a batch -- the first ``counters[0]`` entries (a device-side count, gnnlm_group_assign), -1 after them -- so every row is
        requested from its owner once however many tokens retrieved it.  -> (codes uint8 [P, M], valid uint8 [n' * n_g],
        index int32 [n' * n_g]): slot c of group g lives in row index[g * n_g + c].  Exact mode reads the count (its exchange
        synchronises for the split sizes anyway) and requests exactly that many groups; padded mode stays sync-free."""
        n_g, st, W = 1 + left + right, self.store, self.shard.world
        M = st.codes.shape[1]
        halo = self.shard.halo_left >= left and self.shard.halo_right >= right
        if self.mode == "exact":
            centres = centres[:int(counters[0].item())]
            if not halo or n_g == 1:
                return self.fetch_codes(centres.view(-1, 1), left, right, False)
            return self._fetch_groups(centres, left, right)
        if not halo or n_g == 1:
            raise NotImplementedError("padded exchange of merged groups needs halo shards (Shard(halo_left, halo_right))")
        worst = self.fixed_requests or centres.numel()
        return self._fetch_groups(centres, left, right, cap_requests=min(worst, self.group_requests) if self.group_requests else worst)

    def _gather_codes(self, rows):
        st = self.store
        return self.ops.pq_gather_decode(st.codes, st.centroids, rows, 0, 0, n_store=st.n_store, row0=st.row0,
                                         want_x=False, want_codes=True, want_valid=False)["codes"]

    def _gather_vals(self, rows):
        st = self.store
        return self.ops.pq_gather_decode(st.codes, st.centroids, rows, 0, 0, n_store=st.n_store, row0=st.row0,
                                         vals=st.vals, want_x=False, want_labels=True, want_valid=False)["labels"]

    def fetch_codes(self, ids, left, right, centres_only):
        """-> (fetched_codes uint8 [S, M] in bucketed order, fetched_valid uint8 [S], fetched_index int32 [S])
        for the slots of ``ids`` [n, kg]: slot s lives in row ``fetched_index[s]``."""
        if not centres_only and (left or right) and self.shard.halo_left >= left and self.shard.halo_right >= right:
            return self._fetch_groups(ids, left, right)
        rows = ids.reshape(-1) if centres_only else slot_rows(ids, left, right, self.store.n_store)
        rows = torch.where((rows >= 0) & (rows < self.store.n_store), rows, torch.full_like(rows, -1))
        if self.mode == "padded":
            req = self.fixed_requests * (1 if centres_only else 1 + left + right) if self.fixed_requests else None
            codes, index, cap = self._padded(rows, self._gather_codes, self.store.codes.shape[1], req)
            return codes, ((rows >= 0) & (index.long() < self.shard.world * cap)).to(torch.uint8), index
        self._account_exact(rows.numel(), self.store.codes.shape[1])
        codes, index = exchange_fetch(rows, self.shard, self._gather_codes, self.group, bucket=bucket_hip, unpermute=False)
        return codes, (rows >= 0).to(torch.uint8), index

    def fetch_knn_vals(self, knn_ids):
        """vals[knn_ids] with numpy's negative-index wrap for the -1 padding (knn_model.py:198)."""
        rows = torch.where(knn_ids < 0, knn_ids + self.store.n_store, knn_ids).reshape(-1)
        if self.mode == "padded":
            back, index, _ = self._padded(rows, self._gather_vals, 4)
            back[-1] = -1                                    # rows outside the store never match a target
            return back[index.long()].reshape(knn_ids.shape)
        self._account_exact(rows.numel(), 4)
        return exchange_fetch(rows, self.shard, self._gather_vals, self.group, bucket=bucket_hip).reshape(knn_ids.shape)


class _MapWatchdog:
    """Fail fast instead of hanging while the peers' shards are mapped: `hipIpcOpenMemHandle` of a multi-GB shard has been seen
    never to return (two processes on ONE device), inside a C call no Python timeout can interrupt.  A
    helper CHILD process (plain Python, never touches the GPU; a child, not an exec of this GPU-initialised process) sleeps
    `seconds`, then says why on stderr and kills this rank: the launcher sees a non-zero exit and takes the other ranks down.
    Leaving the block in time stops the helper.  GNNLM_PEER_MAP_TIMEOUT (seconds, default 120; 0: no watchdog)."""

    def __init__(self, what, seconds=None):
        import os
        self.what = what
        self.seconds = float(os.environ.get("GNNLM_PEER_MAP_TIMEOUT", "120")) if seconds is None else float(seconds)
        self.proc = None

    def __enter__(self):
        if self.seconds > 0:
            import os
            import subprocess
            import sys
            code = ("import os, signal, sys, time\n"
                    "pid, secs, what = int(sys.argv[1]), float(sys.argv[2]), sys.argv[3]\n"
                    "t0 = time.time()\n"
                    "while time.time() - t0 < secs:\n"
                    "    time.sleep(0.2)\n"
                    "    if os.getppid() != pid:\n"
                    "        sys.exit(0)\n"
                    "sys.stderr.write('gnnlm_amd.dist: %s did not finish within %.0f s -- the peer-mapped store (--exchange peer) is not usable on this "
                    "transport; use --exchange padded|exact.  Killing rank process %d.\\n' % (what, secs, pid))\n"
                    "sys.stderr.flush()\n"
                    "os.kill(pid, signal.SIGKILL)\n")
            self.proc = subprocess.Popen([sys.executable, "-c", code, str(os.getpid()), str(self.seconds), self.what],
                                         stdin=subprocess.DEVNULL, close_fds=True)
        return self

    def __exit__(self, *exc):
        if self.proc is not None:
            self.proc.kill()
            self.proc.wait()
        return False


class PeerMappedFetcher:
    """The alternative to the exchange: every rank maps its peers' shards into its own address space (HIP
    IPC handles, passed round once with all_gather_object) and gathers the rows it needs itself -- ONE kernel
    (`gnnlm_gather_rows_peer`), no collective, no bucketing, no padding; over xGMI its loads go straight to the owner's
    HBM.  Same contract as :class:`ShardedFetcher` (codes in request order: the index is the identity)."""

    def __init__(self, store, shard: Shard, group=None, share_vals: bool = False):
        from torch.multiprocessing.reductions import reduce_tensor
        from . import _lib
        self._lib, self.store, self.shard, self.group = _lib, store, shard, group
        assert store.row0 == shard.store_row0 and store.codes.shape[0] == shard.store_rows
        W = shard.world
        mine = {"codes": reduce_tensor(store.codes) if store.codes.numel() else None,
                "vals": reduce_tensor(store.vals) if share_vals and store.vals is not None and store.vals.numel() else None,
                "device": store.codes.device.index}
        everyone = [None] * W
        if W > 1:
            dist.all_gather_object(everyone, mine, group=group)
        else:
            everyone[0] = mine
        open_ = lambda h: None if h is None else h[0](*h[1])
        # (seen on the one-GPU test transport only: two processes on ONE device mapping each other's 6.6-GB shards never return from
        # hipIpcOpenMemHandle -- one at a time or both at once; 256-MB shards map at once.  Separate devices are the product's case.)
        # -> under a watchdog: the rank exits non-zero with a message instead of hanging (_MapWatchdog)
        with _MapWatchdog(f"mapping the peers' shards into rank {shard.rank} (hipIpcOpenMemHandle)"):
            self.peers = {k: [getattr(store, k) if g == shard.rank else open_(everyone[g][k]) for g in range(W)] for k in ("codes", "vals")}
            if store.codes.is_cuda:
                torch.cuda.synchronize(store.codes.device)
        here = store.codes.device.index
        for g in range(W):
            if g != shard.rank and everyone[g]["device"] != here:
                _lib.call("gnnlm_enable_peer_access", everyone[g]["device"])
        self._shards = [Shard(shard.n_store, W, g, shard.halo_left, shard.halo_right) for g in range(W)]
        self.share_vals = share_vals
        self.link_bytes = 0

    def check(self):
        pass                                                    # nothing can overflow

    def mapped_store(self):
        """The store with its shard table: handed to the engine, the star-attention / gather kernels read every code row from
        its owner's memory themselves -- no fetch step at all for the codes (``fetch_codes`` stays for consumers that want a
        local copy)."""
        import dataclasses
        return dataclasses.replace(self.store, shards=[(t, sh.store_row0) for t, sh in zip(self.peers["codes"], self._shards)],
                                   rows_per_rank=self.shard.per)

    def _gather(self, rows, what, row_bytes, out):
        L = self._lib
        d = L.gnnlm_peer_gather_t()
        for g, (sh, t) in enumerate(zip(self._shards, self.peers[what])):
            d.shard[g] = t.data_ptr() if t is not None and t.numel() else None
            d.shard_row0[g], d.shard_rows[g] = sh.store_row0, (t.shape[0] if t is not None else 0)
        d.world, d.row_bytes = self.shard.world, row_bytes
        d.rows_per_rank, d.n_store = self.shard.per, self.shard.n_store
        rows = rows.reshape(-1).contiguous()
        valid = torch.empty(rows.numel(), dtype=torch.uint8, device=rows.device)
        d.rows, d.n, d.out, d.out_valid = rows.data_ptr(), rows.numel(), out.data_ptr(), valid.data_ptr()
        L.call_desc("gnnlm_gather_rows_peer", d)
        W = self.shard.world
        self.link_bytes += int(rows.numel() * (W - 1) / W) * row_bytes          # uniform-id expectation, payload only
        return valid

    def fetch_codes(self, ids, left, right, centres_only):
        rows = ids.reshape(-1) if centres_only else slot_rows(ids, left, right, self.store.n_store)
        M = self.store.codes.shape[1]
        codes = torch.empty(rows.numel(), M, dtype=torch.uint8, device=ids.device)
        valid = self._gather(rows, "codes", M, codes)
        return codes, valid, torch.arange(rows.numel(), dtype=torch.int32, device=ids.device)

    def fetch_groups(self, centres, left, right, counters):
        """Merged groups (see ShardedFetcher.fetch_groups); consumers that hold the mapped store (`mapped_store`) never call this."""
        return self.fetch_codes(centres.view(-1, 1), left, right, False)

    def fetch_knn_vals(self, knn_ids):
        assert self.share_vals, "built without the label shards"
        rows = torch.where(knn_ids < 0, knn_ids + self.store.n_store, knn_ids).reshape(-1)
        v = self.store.vals
        out = torch.empty(rows.numel(), dtype=v.dtype, device=knn_ids.device)
        self._gather(rows, "vals", v.element_size(), out)
        return out.to(torch.int32).reshape(knn_ids.shape)
