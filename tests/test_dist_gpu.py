"""The sharded-store exchange over RCCL on one GPU (world size 1): routing + HIP owner-side gather +
fetched-codes HGT path must reproduce the direct path bit for bit (tools/exchange_check.py).  Run in a
subprocess so the process group does not outlive the test."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bucket_rows_kernel():
    """HIP bucketing == the backend-agnostic torch bucketing up to the order inside a bucket."""
    import numpy as np
    import torch
    from gnnlm_amd.dist import Shard, bucket_hip, bucket_torch
    dev = torch.device("cuda:0")
    for world, rank, n_store, n in [(8, 3, 103227021, 200000), (1, 0, 1000, 5000), (3, 2, 50, 1000), (8, 0, 1000, 0)]:
        rs = np.random.RandomState(world + n)
        rows = rs.randint(-1, n_store + 2, size=n).astype(np.int64)
        shard = Shard(n_store, world, rank)
        r = torch.from_numpy(rows).to(dev)
        c1, s1, i1 = bucket_hip(r, shard)
        c0, s0, i0 = bucket_torch(r, shard)
        torch.cuda.synchronize()
        assert torch.equal(c1, c0)
        assert torch.equal(s1[i1.long()], r)                                  # index answers every request
        off = np.concatenate([[0], np.cumsum(c0.cpu().numpy())])
        own = shard.owner(s1).cpu().numpy()
        for o in range(world):                                                # every bucket holds only its owner's rows
            assert (own[off[o]:off[o + 1]] == o).all()
        assert sorted(i1.cpu().tolist()) == list(range(n))                    # a permutation


def test_exchange_over_rccl_single_rank():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exchange_check.py")], capture_output=True,
                       text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("exchange path == direct path") == 2
