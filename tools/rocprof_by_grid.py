#!/usr/bin/env python
"""rocprofv3 --kernel-trace result (rocpd sqlite .db) grouped by (kernel, grid, workgroup): calls, mean / total duration — tells the
call sites of one kernel apart (e.g. which convolutions run on the register-staged k_conv_f16x3).
   python tools/rocprof_by_grid.py x_results.db [name filter] [clips]"""
import re
import sqlite3
import sys

db = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
clips = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
gx = "grid_x" if "grid_x" in cols else "grid_size_x"
wx = "workgroup_x" if "workgroup_x" in cols else "workgroup_size_x"
rows = c.execute(f"select name, {gx}, {wx}, count(*), sum(end-start), avg(end-start), min(end-start) from kernels "
                 f"group by name, {gx}, {wx} order by 5 desc").fetchall()
print(f"{'ms/clip':>9} {'calls/clip':>10} {'avg_us':>9} {'min_us':>9} {'grid':>9} {'wg':>5}  kernel")
for name, g, w, n, s, a, mn in rows:
    short = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "")).replace("void ", "").replace("sampt::", "")
    if flt and flt not in short:
        continue
    print(f"{s / 1e6 / clips:9.3f} {n / clips:10.1f} {a / 1e3:9.1f} {mn / 1e3:9.1f} {g:9d} {w:5d}  {short[:90]}")
