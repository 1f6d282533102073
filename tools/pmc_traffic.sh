#!/bin/bash
# HBM traffic of the gate-GEMM launches: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) + L2 hit counters.
# Run on the GPU box from the repo root: bash tools/pmc_traffic.sh ; then python tools/pmc_traffic.py gpurun_out/pmc
set -e
out=$PWD/gpurun_out/pmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace -d $out/$tag -o pmc -- python $OLDPWD/bench.py --steps 128 --warmup 16 --reps 1 --no-cpu-baseline --no-variants > $out/$tag.log 2>&1 || echo "pass $tag failed"
done
ls -R $out | head -30
