#!/bin/bash
# Is it the clock? GRBM_GUI_ACTIVE (cycles of the graphics clock while the GPU is busy) per launch / launch duration, for the same item with
# 1x and 4x as many CUs busy (tools/lds_load_probe.sh). Two passes each: --pmc (cycles) and --kernel-trace (durations); launches are matched by order.
export TMPDIR=/tmp
for n in 1 4; do
  rm -rf /tmp/cp_$n
  (cd /tmp && RC_DBG_REPLICATE=$n rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d /tmp/cp_$n -o pmc -- python $GRAFT_REPO_ROOT/tools/lds_item_probe.py > /tmp/cp_$n.log 2>&1)
  python - $n <<'PY'
import sqlite3, glob, sys, collections
n = sys.argv[1]
db = glob.glob(f"/tmp/cp_{n}/**/*.db", recursive=True)[0]
con = sqlite3.connect(db); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table' or type='view'")]
disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
scol = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
name_col = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else "name")
rows = cur.execute(f"select d.id, s.{name_col}, d.start, d.end, d.grid_size_x, d.workgroup_size_x from {disp} d join {sym} s on d.kernel_id = s.id order by d.start").fetchall()
pmc = next((t for t in tabs if t.startswith("rocpd_pmc_event")), None)
vals = collections.defaultdict(float)
if pmc:
    cols = [r[1] for r in cur.execute(f"pragma table_info({pmc})")]
    key = "event_id" if "event_id" in cols else cols[1]
    for r in cur.execute(f"select {key}, value from {pmc}"):
        vals[r[0]] += r[1]
else:
    for k, c, v, did in cur.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
        vals[did] += v
ev = {}
try:
    for did, eid in cur.execute(f"select id, event_id from {disp}"): ev[did] = eid
except Exception:
    pass
agg = collections.defaultdict(list)
for did, name, a, b, g, w in rows:
    if "lds_kernel" not in name: continue
    cyc = vals.get(ev.get(did, did), vals.get(did, 0.0))
    agg[g // max(w, 1)].append((cyc, (b - a) / 1e3))
for wg, v in sorted(agg.items()):
    v = v[len(v) // 2:]
    cyc = sum(x[0] for x in v) / len(v); us = sum(x[1] for x in v) / len(v)
    print(f"replicate {n}: {wg:4d} wg  avg {us:7.2f} us  GRBM_GUI_ACTIVE {cyc:12.0f} per launch  -> {cyc / us / 1e3:6.2f} GHz x instances")
PY
done
