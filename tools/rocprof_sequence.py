#!/usr/bin/env python
"""Dump the last N kernel dispatches of a rocprofv3 --kernel-trace result (rocpd sqlite .db) in launch order with their
durations and the gap to the previous kernel's end:  python tools/rocprof_sequence.py x_results.db [N]"""
import re
import sqlite3
import sys

db, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 400
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
extra = [k for k in ("grid_x", "workgroup_x", "grid_size_x", "workgroup_size_x") if k in cols]
rows = c.execute(f"select name, start, end{''.join(', ' + k for k in extra)} from kernels order by start").fetchall()[-n:]
prev = rows[0][1]
for r in rows:
    name = re.sub(r"\(.*", "", r[0].replace("(anonymous namespace)::", "")).replace("void ", "").replace("sampt::", "")
    print(f"{(r[2] - r[1]) / 1e3:9.1f} us  gap {(r[1] - prev) / 1e3:7.1f}  {name[:70]}  {' '.join(str(v) for v in r[3:])}")
    prev = r[2]
