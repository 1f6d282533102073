#!/usr/bin/env python3
"""Idle gaps between consecutive kernel dispatches (all queues merged) in a rocprofv3 results database: the largest gaps of the
last `window_ms` of the trace with the kernels on either side.    python tools/rocpd_gaps.py x_results.db [window_ms=30] [top=25 | -min_gap_us: in time order]"""
import sqlite3, sys

def main(path, window_ms=30.0, top=25):
    db = sqlite3.connect(path); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    scol = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
    name_col = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else "name")
    rows = cur.execute(f"select s.{name_col}, d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id order by d.start").fetchall()
    t_hi = rows[-1][2]
    rows = [r for r in rows if r[1] >= t_hi - window_ms * 1e6]
    gaps, end = [], rows[0][2]
    for k in range(1, len(rows)):
        if rows[k][1] > end:
            gaps.append((rows[k][1] - end, k))
        end = max(end, rows[k][2])
    span = rows[-1][2] - rows[0][1]
    idle = sum(g for g, _ in gaps)
    print(f"{len(rows)} dispatches in the last {span / 1e6:.2f} ms; idle {idle / 1e3:.1f} us in {len(gaps)} gaps")
    short = lambda n: n.split("(")[0][:48]
    order = sorted(gaps, reverse=True)[:top]
    if top < 0:                       # negative `top`: every gap of at least -top us, in time order
        order = [(g, k) for g, k in gaps if g >= -top * 1e3]
    for g, k in order:
        print(f"  gap {g / 1e3:8.1f} us at +{(rows[k][1] - rows[0][1]) / 1e3:9.1f} us   {short(rows[k - 1][0])}  ->  {short(rows[k][0])}")

if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 30.0, int(sys.argv[3]) if len(sys.argv) > 3 else 25)
