#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace result (rocpd sqlite .db) into a per-kernel table (the `--stats` view):
   python tools/rocprof_summary.py gpurun_out/prof/x_results.db [frames] > profiles/<name>.txt"""
import re
import sqlite3
import sys


def main():
    db = sys.argv[1]
    frames = float(sys.argv[2]) if len(sys.argv) > 2 else None
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"# source: {db}")
    print(f"# total kernel time {tot / 1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches"
          + (f"; {tot / 1e6 / frames:.3f} ms/frame over {frames:.0f} frames" if frames else ""))
    print(f"{'total_ms':>10} {'pct':>6} {'calls':>8} {'avg_us':>10} {'min_us':>9} {'max_us':>9}  kernel")
    for name, n, s, a, mn, mx in rows:
        short = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", ""))
        print(f"{s / 1e6:10.2f} {100 * s / tot:6.2f} {n:8d} {a / 1e3:10.1f} {mn / 1e3:9.1f} {mx / 1e3:9.1f}  {short}")


if __name__ == "__main__":
    main()
