#!/bin/bash
# Where the waves of the gate-GEMM kernels spend their cycles (one 8-counter SQ pass) -- input for the next tuning round.
set -e
out=$PWD/gpurun_out/pmc_stall
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VMEM_TA_ADDR_FIFO_FULL \
  --kernel-trace -d $out/a -o pmc -- python $OLDPWD/bench.py --steps 32 --warmup 8 --no-cpu-baseline > $out/a.log 2>&1 || echo "pass a failed"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_IFETCH \
  --kernel-trace -d $out/b -o pmc -- python $OLDPWD/bench.py --steps 32 --warmup 8 --no-cpu-baseline > $out/b.log 2>&1 || echo "pass b failed"
cd $OLDPWD
python - <<'PY'
import sqlite3, glob, collections, json
res = collections.defaultdict(dict)
for tag in ("a", "b"):
    dbs = glob.glob(f"gpurun_out/pmc_stall/{tag}/**/*.db", recursive=True)
    if not dbs:
        print(tag, "no db"); continue
    con = sqlite3.connect(dbs[0])
    rows = list(con.execute("select kernel_name, counter_name, value, grid_size from counters_collection"))
    agg = collections.defaultdict(list)
    for k, c, v, g in rows:
        if k.startswith("rc_gemm"):
            agg[(k.split("(")[0] + ":%d" % (g // 256), c)].append(v)
    for (k, c), v in sorted(agg.items()):
        res[k][c] = sum(v) / len(v)
print(json.dumps(res, indent=1))
json.dump(res, open("gpurun_out/pmc_stall/summary.json", "w"), indent=1)
PY
