"""The sharded-store exchange over RCCL on one GPU (world size 1): routing + HIP owner-side gather +
fetched-codes HGT path must reproduce the direct path bit for bit (tools/exchange_check.py).  Run in a
subprocess so the process group does not outlive the test."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exchange_over_rccl_single_rank():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exchange_check.py")], capture_output=True,
                       text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("exchange path == direct path") == 2
