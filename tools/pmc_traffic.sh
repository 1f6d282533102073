#!/bin/bash
# HBM traffic of the gate-GEMM launches: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) + L2 hit counters.
# Run on the GPU box from the repo root: [STEPS=128] bash tools/pmc_traffic.sh ; then python tools/pmc_traffic.py gpurun_out/pmc$STEPS 256 mixed $STEPS
# (the launch population -- launches per step, FLOPs per launch -- depends on the frames per call: one record per --steps value)
set -e
STEPS=${STEPS:-128}
out=$PWD/gpurun_out/pmc$STEPS
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace -d $out/$tag -o pmc -- python $OLDPWD/bench.py --steps $STEPS --warmup ${WARMUP:-16} --reps 1 --no-cpu-baseline --no-variants > $out/$tag.log 2>&1 || echo "pass $tag failed"
done
ls -R $out | head -30
