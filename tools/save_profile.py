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
# the search's kernels (tools/profile_search.sh <tag>, if it ran): FETCH_SIZE / WRITE_SIZE in KB per launch, the same corrections
stxt = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_search.txt")
if os.path.exists(stxt):
    import ast
    import json
    import re
    acc = {}
    for l in open(stxt):
        m = re.match(r"pmc_ivf8([fw]) (\S+?)[(,<]?\S* (\{.*\})\s*$", l) if l.startswith("pmc_ivf8f") or l.startswith("pmc_ivf8w") else None
        if not m:
            continue
        name = l.split(" ", 2)[1]
        vals = ast.literal_eval(m.group(3))
        key = "search:" + ("ivfpq_scan8_kernel<false>" if name.startswith("ivfpq_scan8_kernel<false>") else
                           "ivfpq_scan8_kernel<true>" if name.startswith("ivfpq_scan8_kernel<true>") else
                           "ivfpq_rescore_kernel" if name.startswith("ivfpq_rescore") else
                           "ivfpq_tau_kernel" if name.startswith("ivfpq_tau") else "topk_kernel")
        e = acc.setdefault(key, {"launches": 1, "fetch_size_bytes_raw": 0.0, "write_size_bytes_raw": 0.0})
        if "FETCH_SIZE" in vals:
            e["fetch_size_bytes_raw"] = float(vals["FETCH_SIZE"]) * 1024
        if "WRITE_SIZE" in vals:
            e["write_size_bytes_raw"] = float(vals["WRITE_SIZE"]) * 1024
    pj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    t = json.load(open(pj))
    for k, e in acc.items():
        e["hbm_bytes_per_launch"] = 2 * e["fetch_size_bytes_raw"] + e["write_size_bytes_raw"]
        t[k] = e
    json.dump(t, open(pj, "w"), indent=1, sort_keys=True)
# the 3-layer recipe's own passes (tools/profile_bench.sh writes prof_<tag>_L3): kernels under "L3:<name>" in pmc_traffic.json
src3 = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_L3")
st3 = glob.glob(os.path.join(src3, "stats", "**", "*_kernel_stats.csv"), recursive=True)
if st3:
    import json
    with open(st3[0]) as f, open(os.path.join(ROOT, "profiles", f"{tag}_L3_kernel_stats.csv"), "w", newline="") as g:
        w = csv.writer(g, quoting=csv.QUOTE_MINIMAL)
        for row in csv.reader(f):
            row[0] = row[0][:160]
            w.writerow(row)
    py3 = [sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), f"FETCH_SIZE={src3}/fetch", f"WRITE_SIZE={src3}/write"]
    open(os.path.join(ROOT, "profiles", f"{tag}_L3_pmc_traffic.csv"), "w").write(subprocess.run(py3, capture_output=True, text=True, check=True).stdout)
    t3 = json.loads(subprocess.run(py3 + ["--json", f"--profile={tag}"], capture_output=True, text=True, check=True).stdout)
    pj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    t = json.load(open(pj))
    for k, e in t3.items():
        if k != "_meta":
            t["L3:" + k] = e
    json.dump(t, open(pj, "w"), indent=1, sort_keys=True)
    l3 = [l for l in open(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_L3.bench.json")) if l.startswith("{")]
    if l3:
        open(os.path.join(ROOT, "profiles", f"{tag}_L3_bench_under_rocprof.json"), "w").write(l3[0])
line = [l for l in open(os.path.join(ROOT, "gpurun_out", f"prof_{tag}.bench.json")) if l.startswith("{")][0]
open(os.path.join(ROOT, "profiles", f"{tag}_bench_under_rocprof.json"), "w").write(line)
print("saved profiles/", tag)
