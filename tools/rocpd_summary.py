#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, min, max duration) of a rocprofv3 run.

rocprofv3 on ROCm 7.2 writes a rocpd SQLite database (`*_results.db`) by default; this prints the
same table `--stats` would give as CSV so it can be committed under profiles/.
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.csv
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name} order by sum(end-start) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
    for n, c, s, a, mn, mx in rows:
        print(f"\"{n}\",{c},{s},{a:.1f},{100.0 * s / tot:.2f},{mn},{mx}")


if __name__ == "__main__":
    main(sys.argv[1])
