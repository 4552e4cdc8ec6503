"""Oracle: the token/neighbour heterograph (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Restates ``GraphTokenBlockDataset.new_build_graph`` and its static helpers
(fairseq/data/token_block_dataset.py:338-412, :545-594) without DGL: the graph
is returned as plain index arrays.

Reference bug kept visible, not silently fixed:
``token_block_dataset.py:384`` bounds the right context with
``len(self.neighbor_offsets.shape[0])`` which raises ``TypeError`` whenever the
right context is > 0.  The intended bound is the neighbour-corpus length; here
it is the explicit ``n_store`` argument.
"""
from typing import Dict, List, Tuple

import numpy as np


def build_ntgt_edges(offsets2id: Dict[int, int], context: int = 0, bidirect: bool = False
                     ) -> Tuple[List[int], List[int]]:
    """Edges between ntgt nodes whose offsets differ by <= context.

    Restates token_block_dataset.py:545-584 (two-pointer sweep over nodes sorted
    by offset; every pair (s <= e) inside the window gives s->e, mirrored when
    ``bidirect``)."""
    if not offsets2id:
        return [], []
    nodes = sorted(((nid, off) for off, nid in offsets2id.items()), key=lambda x: x[1])
    src, tgt = [], []
    start, end, length = 0, -1, len(nodes)
    while start < length:
        while end + 1 < length and nodes[end + 1][1] <= nodes[start][1] + context:
            end += 1
            for s in range(start, end + 1):
                src.append(nodes[s][0])
                tgt.append(nodes[end][0])
        start += 1
    if bidirect:
        for idx in range(len(src)):
            s, t = src[idx], tgt[idx]
            if s != t:
                src.append(t)
                tgt.append(s)
    return src, tgt


def auto_regressive_edges(length: int, max_context: int = 0):
    """tgt u -> tgt w for all u <= w (and w-u < max_context if given).

    Restates token_block_dataset.py:586-594 (torch.triu mask -> torch.where,
    row-major order)."""
    mask = np.triu(np.ones((length, length), dtype=bool))
    if max_context:
        mask &= ~np.triu(np.ones((length, length), dtype=bool), k=max_context)
    us, vs = np.nonzero(mask)
    return us.astype(np.int64), vs.astype(np.int64)


def build_graph(neighbor_idxs, tgt_offsets, n_store, left, right,
                invalid_neighbor_context=0, max_intra_context=0):
    """Reference-order graph of one block.  Restates token_block_dataset.py:338-412.

    Args:
        neighbor_idxs: int64 [T, k]  rows of neighbors.mmap for the block's tokens (-1 = none)
        tgt_offsets:   int64 [T]     global offsets of the block's tokens (only used by the
                                     invalid_neighbor_context filter, :361; 0 at eval)
        n_store: neighbour-corpus length (bound of the right context, see module docstring)
        left/right: context sizes (--neighbor-context)
    Returns dict with
        ntgt_offsets [N_ntgt] int64  datastore row of every ntgt node, reference order
        ntgt_group   [N_ntgt] int64  index of the (token, neighbour) group the node belongs to
        group_tgt    [G]      int64  token index of each group
        inter        (src ntgt ids, dst tgt ids)           ('ntgt','inter','tgt')
        intra_ntgt   (src ntgt ids, dst ntgt ids)          ('ntgt','intra','ntgt')
        intra_tgt    (src tgt ids, dst tgt ids)            ('tgt','intra','tgt')
    """
    neighbor_idxs = np.asarray(neighbor_idxs)
    T = neighbor_idxs.shape[0]
    ntgt_id = 0
    tgt2ntgt = [[], []]
    ntgt2ntgt = [[], []]
    ntgt_offsets, ntgt_group, group_tgt = [], [], []
    for tgt_idx in range(T):
        for offset in neighbor_idxs[tgt_idx]:
            offset = int(offset)
            if offset == -1:                                              # :358
                continue
            if abs(int(tgt_offsets[tgt_idx]) - offset) < invalid_neighbor_context:   # :361
                continue
            g = len(group_tgt)
            group_tgt.append(tgt_idx)
            cur_ids, cur_offsets = [ntgt_id], [offset]
            ntgt_offsets.append(offset)
            ntgt_group.append(g)
            tgt2ntgt[0].append(tgt_idx)
            tgt2ntgt[1].append(ntgt_id)
            ntgt_id += 1
            context_offsets = []
            if left:
                context_offsets.extend(range(max(0, offset - left), offset))            # :380-381
            if right:
                context_offsets.extend(range(offset + 1, min(n_store, offset + 1 + right)))  # :384
            for o in context_offsets:
                cur_ids.append(ntgt_id)
                cur_offsets.append(o)
                ntgt_offsets.append(o)
                ntgt_group.append(g)
                ntgt_id += 1
            s, t = build_ntgt_edges({o: i for i, o in zip(cur_ids, cur_offsets)},
                                    context=1, bidirect=True)            # :395-398
            ntgt2ntgt[0].extend(s)
            ntgt2ntgt[1].extend(t)
    us, vs = auto_regressive_edges(T, max_intra_context)
    i64 = lambda a: np.asarray(a, dtype=np.int64)
    return {
        "num_tgt": T,
        "num_ntgt": ntgt_id,
        "ntgt_offsets": i64(ntgt_offsets),
        "ntgt_group": i64(ntgt_group),
        "group_tgt": i64(group_tgt),
        "inter": (i64(tgt2ntgt[1]), i64(tgt2ntgt[0])),
        "intra_ntgt": (i64(ntgt2ntgt[0]), i64(ntgt2ntgt[1])),
        "intra_tgt": (us, vs),
    }


def slot_layout(neighbor_idxs, n_store, left, right, tgt_offsets=None, invalid_neighbor_context=0):
    """Vectorised padded-slot view of the same graph (what the HIP path uses).

    Slot ``(i, j, c)`` with ``c = 0`` the centre ``o = nb[i, j]``, ``c = 1..left`` the rows
    ``o-left .. o-1`` and ``c = left+1 .. left+right`` the rows ``o+1 .. o+right``.  A slot is
    valid iff its group is valid (``o != -1`` and, with ``invalid_neighbor_context = c > 0``,
    ``|tgt_offsets[i] - o| >= c``: token_block_dataset.py:358-362) and its row lies in ``[0, n_store)``.  Walking
    the valid slots in (i, j, c) order reproduces the reference node order of
    :func:`build_graph` exactly.

    Returns (rows int64 [T,k,n_g] with -1 for invalid slots, valid bool [T,k,n_g])."""
    nb = np.asarray(neighbor_idxs, dtype=np.int64)
    n_g = 1 + left + right
    delta = np.concatenate([[0], np.arange(-left, 0), np.arange(1, right + 1)]).astype(np.int64)
    rows = nb[:, :, None] + delta[None, None, :]
    group_ok = nb != -1
    if invalid_neighbor_context > 0:
        group_ok &= np.abs(np.asarray(tgt_offsets, dtype=np.int64)[:, None] - nb) >= invalid_neighbor_context
    valid = group_ok[:, :, None] & (rows >= 0) & (rows < n_store)
    rows = np.where(valid, rows, -1)
    assert rows.shape[-1] == n_g
    return rows, valid
