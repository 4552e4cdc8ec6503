#!/usr/bin/env python3
"""Copy the judged summaries of a tools/profile_bench.sh run from gpurun_out/ (scratch) into profiles/ (tracked):
    python tools/save_profile.py r02a
-> profiles/<tag>_kernel_stats.csv (rocprofv3 --stats, kernel names cut to 160 characters), <tag>_pmc_traffic.csv,
   <tag>_bench_under_rocprof.json, and profiles/pmc_traffic.json stamped with the kernel sources it was measured on."""
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
stats = glob.glob(os.path.join(src, "stats", "**", "*_kernel_stats.csv"), recursive=True)[0]
with open(stats) as f, open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"), "w", newline="") as g:
    w = csv.writer(g, quoting=csv.QUOTE_MINIMAL)
    for row in csv.reader(f):
        row[0] = row[0][:160]
        w.writerow(row)
py = [sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), f"FETCH_SIZE={src}/fetch", f"WRITE_SIZE={src}/write"]
open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.csv"), "w").write(subprocess.run(py, capture_output=True, text=True, check=True).stdout)
open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w").write(
    subprocess.run(py + ["--json", f"--profile={tag}"], capture_output=True, text=True, check=True).stdout)
line = [l for l in open(os.path.join(ROOT, "gpurun_out", f"prof_{tag}.bench.json")) if l.startswith("{")][0]
open(os.path.join(ROOT, "profiles", f"{tag}_bench_under_rocprof.json"), "w").write(line)
print("saved profiles/", tag)
