"""bench.py's launcher half, which needs no GPU: a plain `python bench.py --gpus N` must start its own ranks (or refuse
with a message when the GPUs are not there) instead of asserting on WORLD_SIZE."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    return {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                             "GNNLM_BENCH_BACKEND")}


def test_plain_multi_gpu_launch_is_refused_with_a_message_when_gpus_are_missing():
    import torch
    n = torch.cuda.device_count() + 2
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--small"], capture_output=True,
                       text=True, timeout=300, env=_env(), cwd=ROOT)
    assert p.returncode == 2, p.stderr[-2000:]
    assert "GPU(s) are visible" in p.stderr and "AssertionError" not in p.stderr and "Traceback" not in p.stderr


def test_plain_multi_gpu_launch_spawns_torch_distributed_run(monkeypatch):
    """The command the launcher builds: torch.distributed.run, one process per GPU, loopback rendezvous, the original flags."""
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    seen = {}

    class _Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return _Done()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setenv("GNNLM_BENCH_BACKEND", "gloo")                # (no GPU here: the one-GPU test transport skips the device count)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    rc = bench.launch_ranks(bench.parse())
    assert rc == 7                                                   # the launcher's exit code is relayed
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_roofline_helpers_are_consistent():
    """The filter's roofline line (`bench.filter_roofline`): achieved = table-byte look-ups x 32 int8 ops / time, priced against the dense
    int8 peak; the PMC traffic of a profile id is the launch-weighted mean over the kernels launched under it (`bench.PMC_FAMILIES`:
    the GEMM id covers the log-sum-exp A-stationary kernel, the filter id only the `<false>` instantiation) and is dropped when the
    profile was measured on other kernel sources."""
    import importlib
    import json
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    r = bench.filter_roofline(total_ms=28.0, launches=1, pairs=32768 * 806461.0)
    assert r["bound"] == "mfma" and r["unit"].startswith("TOP/s") and abs(r["achieved"] - 32768 * 806461 * 64 * 32 / 0.028 / 1e12) < 0.1
    assert abs(r["frac"] - r["achieved"] / 5000.0) < 1e-3 and 0.3 < r["frac"] < 0.5
    fams = bench.PMC_FAMILIES
    assert any("gemm_lse_astationary".startswith(f) for f in fams["gemm_nt_f32_kernel"])
    assert "ivfpq_scan8_kernel<false>" in fams["ivfpq_scan8_kernel"] and "ivfpq_scan8_kernel<true>" not in fams["ivfpq_scan8_kernel"]
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
        t = json.load(f)
    stamped = t["_meta"]["kernel_source_hash"] == bench.kernel_source_hash()
    val, src = bench.pmc_traffic("ivfpq_scan8_kernel")
    if stamped:
        k = t["ivfpq_scan8_kernel<false>"]
        assert val == round(k["hbm_bytes_per_launch"]) and src == t["_meta"]["profile"]
    else:
        assert val is None and "other kernel sources" in src
