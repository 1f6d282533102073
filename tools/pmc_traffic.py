#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc databases written by tools/pmc_traffic.sh to fabric-side bytes per launch of the dominant
kernel (the shared-weight kernel of the LSTM layer steps rc_gemm_lds_kernel where it runs, else the wide-tile gate GEMM rc_gemm_split_kernel / rc_gemm_kernel), keyed by the workload
(batch, confidence schedule, frames per call) so that bench.py only quotes it for the run it belongs to.
    python tools/pmc_traffic.py gpurun_out/pmc128 [batch] [conf] [steps] > profiles/rNN_pmc_traffic_steps128.json"""
import glob
import json
import os
import sqlite3
import sys

root = sys.argv[1]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
conf = sys.argv[3] if len(sys.argv) > 3 else "mixed"
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 128


def per_launch(tag, counters):
    db = glob.glob(os.path.join(root, tag, "**", "*.db"), recursive=True)[0]
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tabs else [t for t in tabs if "counter" in t.lower()][0]
    out = {}
    for c in counters:
        rows = list(con.execute(f"select kernel_name, value from {view} where counter_name = ?", (c,)))
        by = {}
        for n, v in rows:
            by.setdefault(n.split("(")[0], []).append(v)
        out[c] = {k: (sum(v) / len(v), len(v)) for k, v in by.items() if k.startswith("rc_gemm")}
    return out


res = {}
for tag, cs in (("FETCH_SIZE", ["FETCH_SIZE"]), ("WRITE_SIZE", ["WRITE_SIZE"]), ("TCC_HIT_sum", ["TCC_HIT_sum", "TCC_MISS_sum"])):
    try:
        res.update(per_launch(tag, cs))
    except Exception as e:  # noqa: BLE001
        print("pass", tag, "failed:", e, file=sys.stderr)
wide = next((k for k in ("rc_gemm_lds_kernel", "rc_gemm_split_kernel", "rc_gemm_kernel") if k in res.get("FETCH_SIZE", {})), None)
if wide and wide in res.get("WRITE_SIZE", {}):
    f, n = res["FETCH_SIZE"][wide]
    w, _ = res["WRITE_SIZE"][wide]
    out = {"batch": batch, "conf": conf, "steps": steps, "kernel": wide,
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum in three separate passes with --kernel-trace only "
                     "(tools/pmc_traffic.sh: bench.py --steps %d --reps 1 --no-cpu-baseline --no-variants), per-launch average over "
                     "%d launches of %s" % (steps, n, wide),
           "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w,
           "traffic_bytes_per_launch": (2 * f + w) * 1024,
           "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncalibrated",
           "other_kernels_KB_per_launch": {k: {"fetch": res["FETCH_SIZE"][k][0], "write": res["WRITE_SIZE"].get(k, (0, 0))[0], "launches": res["FETCH_SIZE"][k][1]}
                                           for k in res["FETCH_SIZE"] if k != wide}}
    if wide in res.get("TCC_HIT_sum", {}):
        h, m = res["TCC_HIT_sum"][wide][0], res["TCC_MISS_sum"][wide][0]
        out["l2_hit_rate"] = h / (h + m)
    print(json.dumps(out, indent=1))
else:
    print(json.dumps({"error": "no wide-tile gate-GEMM launches found", "kernels": sorted(res.get("FETCH_SIZE", {}))}))
    sys.exit(1)
