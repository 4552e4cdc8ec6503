"""Determinism / race screen of the barrier- and LDS-DMA-synchronised kernels (tools/race_screen.py): repeated
launches on identical inputs must be bit-identical."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kernels_are_bit_stable():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "race_screen.py")], capture_output=True, text=True,
                       timeout=900, env=dict(os.environ, REPS="12"), cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "MISMATCH" not in p.stdout and p.stdout.count("OK") >= 10
