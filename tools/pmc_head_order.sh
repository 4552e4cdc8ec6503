#!/bin/bash
# FETCH_SIZE (L2 -> fabric requests, KB; 128-B requests count as 64 B on gfx950: double it) of the head GEMM per tile walk
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$R
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/head_gemm_order.py
for o in ${ORDERS:-0 1 2 6 10 18}; do
  rm -rf $R/gpurun_out/pmc_head_$o
  TILE_ORDERS=$o REPS=3 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_head_$o --output-format csv -- python3 $R/tools/head_gemm_order.py > /dev/null 2>&1
  python3 - $o <<'PY'
import csv, glob, os, sys
R = os.environ["GRAFT_REPO_ROOT"]; o = sys.argv[1]
fs = glob.glob(f"{R}/gpurun_out/pmc_head_{o}/**/*counter_collection.csv", recursive=True)
if not fs:
    print("tile_order", o, "no counter file"); sys.exit(0)
acc = {}
for r in csv.DictReader(open(fs[0])):
    if "gemm_nt" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        acc[r["Dispatch_Id"]] = acc.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
v = sorted(acc.values())
print(f"tile_order {o}: FETCH_SIZE per launch {v[len(v)//2] / 1e6:.3f} GB raw (x2 = {2 * v[len(v)//2] / 1e6:.3f} GB), {len(v)} launches")
PY
done
