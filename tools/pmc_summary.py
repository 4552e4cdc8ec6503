#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes (one counter per pass, CSV output).

    python tools/pmc_summary.py FETCH_SIZE=dir_a WRITE_SIZE=dir_b > profiles/rNN_pmc_traffic.csv
    python tools/pmc_summary.py --json --profile=rNNx ... > profiles/pmc_traffic.json      (read by bench.py)

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB.  `hbm_bytes_per_launch` applies the gfx950 correction
of MI355X_MICROARCH.md (FETCH_SIZE counts 128-B read requests as 64 B: doubled) and adds WRITE_SIZE as is.
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"gnnlm::\(anonymous namespace\)::(\w+(?:<[^>(]*>)?)", name)
    return m.group(1) if m else None


def main():
    as_json = "--json" in sys.argv
    acc = defaultdict(lambda: defaultdict(list))
    for arg in sys.argv[1:]:
        if "=" not in arg or arg.startswith("--"):
            continue
        counter, d = arg.split("=", 1)
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                if k and row["Counter_Name"] == counter:
                    acc[k][counter].append(float(row["Counter_Value"]))
    out = {}
    for k, c in acc.items():
        f = c.get("FETCH_SIZE", [])
        w = c.get("WRITE_SIZE", [])
        fa = sum(f) / len(f) * 1024 if f else None
        wa = sum(w) / len(w) * 1024 if w else None
        out[k] = {"launches": max(len(f), len(w)), "fetch_size_bytes_raw": fa, "write_size_bytes_raw": wa,
                  "hbm_bytes_per_launch": (2 * fa if fa is not None else 0) + (wa or 0)}
    if as_json:
        # stamp with the kernel sources the passes were measured on: bench.py drops the numbers when they are stale
        import importlib.util
        import os
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        name = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--profile=")), "unnamed")
        out["_meta"] = {"kernel_source_hash": bench.kernel_source_hash(), "profile": name}
        print(json.dumps(out, indent=1, sort_keys=True))
        return
    print("kernel,launches,FETCH_SIZE_avg_bytes(raw),WRITE_SIZE_avg_bytes(raw),hbm_bytes_per_launch(2xFETCH+WRITE)")
    for k, v in sorted(out.items(), key=lambda kv: -(kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])):
        print(f"\"{k}\",{v['launches']},{v['fetch_size_bytes_raw'] or 0:.0f},{v['write_size_bytes_raw'] or 0:.0f},{v['hbm_bytes_per_launch']:.0f}")


if __name__ == "__main__":
    main()
