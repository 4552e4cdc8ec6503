#!/usr/bin/env python3
"""Per-dispatch durations (in launch order) of the last N kernel dispatches of a rocprofv3 rocpd database.
    python tools/rocpd_trace.py x_results.db [N]
"""
import sqlite3
import sys


def main(path, n):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    extra = [c for c in ("grid_x", "grid_size_x", "workgroup_x", "workgroup_size_x") if c in cols]
    sel = ", ".join([name, "start", "end"] + extra)
    rows = cur.execute(f"select {sel} from kernels order by start").fetchall()
    for r in rows[-n:]:
        print(f"{(r[2] - r[1]) / 1e3:10.1f} us  {r[0][:70]}  {list(r[3:])}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
