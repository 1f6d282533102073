#!/bin/bash
# MFMA utilisation of the gate-GEMM kernels: rocprofv3 --pmc passes (derived MfmaUtil; raw busy cycles + GRBM_GUI_ACTIVE).
set -e
out=$PWD/gpurun_out/pmc_mfma
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc MfmaUtil --kernel-trace -d $out/util -o pmc -- python $OLDPWD/bench.py --steps 128 --warmup 16 --reps 1 --no-cpu-baseline --no-variants > $out/util.log 2>&1 || echo "pass MfmaUtil failed"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out/raw -o pmc -- python $OLDPWD/bench.py --steps 128 --warmup 16 --reps 1 --no-cpu-baseline --no-variants > $out/raw.log 2>&1 || echo "pass raw failed"
cd $OLDPWD
python - <<'PY'
import sqlite3, glob, collections, json
res = {}
for tag in ("util", "raw"):
    dbs = glob.glob(f"gpurun_out/pmc_mfma/{tag}/**/*.db", recursive=True)
    if not dbs:
        print(tag, "no db"); continue
    con = sqlite3.connect(dbs[0])
    rows = list(con.execute("select kernel_name, counter_name, value from counters_collection"))
    agg = collections.defaultdict(list)
    for k, c, v in rows:
        if k.startswith("rc_gemm"):
            agg[(k.split("(")[0], c)].append(v)
    for (k, c), v in sorted(agg.items()):
        res[f"{k}:{c}"] = {"mean": sum(v) / len(v), "n": len(v)}
print(json.dumps(res, indent=1))
json.dump(res, open("gpurun_out/pmc_mfma/summary.json", "w"), indent=1)
PY
