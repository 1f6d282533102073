#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc databases written by tools/pmc_traffic.sh to bytes per gate-GEMM launch."""
import json, sqlite3, sys, os, glob
root = sys.argv[1]
def per_launch(tag, counters):
    db = glob.glob(os.path.join(root, tag, "**", "*.db"), recursive=True)[0]
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tabs else [t for t in tabs if "counter" in t.lower()][0]
    cols = [d[1] for d in con.execute(f"pragma table_info({view})")]
    out = {}
    for c in counters:
        rows = list(con.execute(f"select kernel_name, value from {view} where counter_name = ?", (c,))) if "kernel_name" in cols else []
        vals = [v for n, v in rows if n.startswith("rc_gemm")]
        out[c] = (sum(vals) / max(len(vals), 1), len(vals))
    return out, cols, view
res = {}
for tag, cs in (("FETCH_SIZE", ["FETCH_SIZE"]), ("WRITE_SIZE", ["WRITE_SIZE"]), ("TCC_HIT_sum", ["TCC_HIT_sum", "TCC_MISS_sum"])):
    try:
        o, cols, view = per_launch(tag, cs)
        res.update(o)
    except Exception as e:
        print("pass", tag, "failed:", e)
print(res)
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    f, n = res["FETCH_SIZE"]; w, _ = res["WRITE_SIZE"]
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc_traffic.sh), bench.py --steps 32 --warmup 8, gate-GEMM launches (rc_gemm_kernel + rc_gemm_small_kernel), per-launch average over %d launches (round-1 final build)" % n,
           "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w,
           "traffic_bytes_per_launch": (2 * f + w) * 1024,
           "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncalibrated",
           "unique_weight_bytes_per_launch": 243.06e6 / 11}
    if "TCC_HIT_sum" in res and res["TCC_HIT_sum"][1]:
        h, m = res["TCC_HIT_sum"][0], res["TCC_MISS_sum"][0]
        out["l2_hit_rate"] = h / (h + m)
    print(json.dumps(out, indent=1))
    json.dump(out, open(os.path.join(root, "pmc_traffic.json"), "w"), indent=1)
