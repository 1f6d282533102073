#!/usr/bin/env python3
"""Kernel timeline of a steady-state window from a rocprofv3 results database (rocpd sqlite): per kernel name and stream the
launches, busy time and the gaps between consecutive kernels of the same stream; plus the union busy time over all streams.
    python tools/rocpd_timeline.py x_results.db [skip_fraction_of_dispatches=0.5] [window_us=2000]"""
import sqlite3, sys, collections

def main(path, skip=0.5, window_us=2000.0):
    db = sqlite3.connect(path); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    scol = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
    name_col = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else "name")
    rows = cur.execute(f"select s.{name_col}, d.stream_id, d.queue_id, d.start, d.end, d.grid_size_x, d.workgroup_size_x from {disp} d "
                       f"join {sym} s on d.kernel_id = s.id order by d.start").fetchall()
    t_lo, t_hi = rows[0][3], rows[-1][4]
    w0 = rows[min(len(rows) - 1, int(len(rows) * skip))][3]                # the window starts at dispatch number skip * N
    w1 = w0 + window_us * 1e3
    win = [r for r in rows if r[3] >= w0 and r[4] <= w1]
    if not win:
        raise SystemExit(f"no dispatch inside the window ({skip:.0%} + {window_us:.0f} us of a {(t_hi - t_lo) / 1e3:.0f} us trace): pick another skip fraction")
    print(f"{len(rows)} dispatches, window {window_us:.0f} us from {skip:.0%}: {len(win)} dispatches")
    # union busy
    busy, cur_s, cur_e = 0, None, None
    for r in win:
        if cur_e is None or r[3] > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = r[3], r[4]
        else:
            cur_e = max(cur_e, r[4])
    busy += cur_e - cur_s
    span = win[-1][4] - win[0][3]
    print(f"span {span / 1e3:.1f} us, some kernel running {busy / 1e3:.1f} us ({100.0 * busy / span:.1f} %), idle {(span - busy) / 1e3:.1f} us")
    per = collections.defaultdict(lambda: [0, 0.0])
    for r in win:
        k = (r[0].split("(")[0], r[2])
        per[k][0] += 1; per[k][1] += (r[4] - r[3]) / 1e3
    for (n, q), (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"  queue {q:>3}  {n:<34} x{c:<5} total {t:9.1f} us  avg {t / c:7.2f} us  ({100.0 * t * 1e3 / span:.1f} % of span)")
    # first ~40 dispatches of the window as a timeline
    print("timeline (us from window start): start  dur  queue  kernel  grid/wg")
    for r in win[:48]:
        print(f"  {(r[3] - win[0][3]) / 1e3:9.2f} {(r[4] - r[3]) / 1e3:8.2f}  q{r[2]}  {r[0].split('(')[0]:<34} {r[5] // max(r[6], 1)} wg")

if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5, float(sys.argv[3]) if len(sys.argv) > 3 else 2000.0)
