#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 results database (rocpd sqlite, the default output format of ROCm 7.2's
`rocprofv3 --kernel-trace --stats`): name, calls, total / average / min / max duration (us), share of GPU time.
    python tools/rocpd_stats.py gpurun_out/.../x_results.db [out.csv]"""
import csv
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in cur.execute(f"pragma table_info({disp})")]
    scol = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
    name_col = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else "name")
    key = "kernel_id" if "kernel_id" in cols else "kernel_symbol_id"
    rows = cur.execute(f"select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
                       f"from {disp} d join {sym} s on d.{key} = s.id group by s.{name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    table = [(n.split("(")[0], c, t / 1e3, t / c / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / total) for n, c, t, mn, mx in rows]
    w = csv.writer(open(out, "w", newline="") if out else sys.stdout)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"])
    for r in table:
        w.writerow([r[0], r[1]] + [f"{v:.2f}" for v in r[2:]])


if __name__ == "__main__":
    main(*sys.argv[1:3])
