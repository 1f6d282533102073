#!/usr/bin/env python3
"""Print the per-kernel timeline of one steady-state frame from a rocprofv3 rocpd database (kernel-trace)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, grid_x, workgroup_x, start, end from kernels order by start"))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
n_show = int(sys.argv[3]) if len(sys.argv) > 3 else 16
seg = rows[len(rows) - back:len(rows) - back + n_show]
t0 = seg[0][3]
for r in seg:
    print("%-24s WGs %5d  start %8.1f us  dur %6.1f us" % (r[0][:24], r[1] // max(r[2], 1), (r[3] - t0) / 1e3, (r[4] - r[3]) / 1e3))
