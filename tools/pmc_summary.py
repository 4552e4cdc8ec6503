#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes (one counter per pass, CSV output).

    python tools/pmc_summary.py FETCH_SIZE=dir_a WRITE_SIZE=dir_b > profiles/rNN_pmc_traffic.csv
    python tools/pmc_summary.py --json --profile=rNNx ... > profiles/pmc_traffic.json      (read by bench.py)

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB.  `hbm_bytes_per_launch` applies the gfx950 correction
of MI355X_MICROARCH.md (FETCH_SIZE counts 128-B read requests as 64 B: doubled) and adds WRITE_SIZE as is.

CALIBRATION (round 4, tools/probes/fetch_calib.hip under tools/pmc_fetch_calib.sh; known byte counts over a 6.4-GB buffer):
    access pattern                                   reported / requested
    stream, 16 B per lane, consecutive lanes          0.50      (the guide's x2 case: 128-B requests counted as 64 B)
    random 128-B rows (8 x 16 B)                      0.50
    random  64-B rows (4 x 16 B)                      1.00
    random   4-B words                               16.0
i.e. FETCH_SIZE = 64 B x (number of read REQUESTS that leave the L2), whatever the request's size up to 128 B.  The x2 is exact
for full-line traffic; for sub-line random reads (the re-score's 64-B code rows, the 4-B label gather of knn_interp) the counter
says how many requests were made, not how many bytes DRAM moved, and 2 x FETCH_SIZE is an UPPER bound of the traffic (a 64-B
request may or may not cost a 128-B burst).  `fetch_requests_per_launch` = raw FETCH_SIZE / 64 is therefore reported beside the
bytes: the memory system serves ~48 G such requests/s (the probe's cooperative 64-B and 128-B row fetches and its 4-B gather all
saturate there), which is what bounds those kernels -- not bytes.
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
                  "hbm_bytes_per_launch": (2 * fa if fa is not None else 0) + (wa or 0),
                  "fetch_requests_per_launch": (fa / 64 if fa is not None else None)}
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
