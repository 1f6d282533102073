set -e
out=$PWD/gpurun_out/tl
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python $OLDPWD/bench.py --steps 128 --warmup 16 --reps 1 --no-cpu-baseline --no-variants $BENCH_ARGS > $out/kt.log 2>&1 || echo "trace failed"
cd $OLDPWD
db=$(find $out/kt -name "*.db" | head -1)
python tools/rocpd_timeline.py $db 0.5 1200 > gpurun_out/r06_timeline_$TAG.txt 2>&1 || true
python tools/rocpd_stats.py $db > gpurun_out/r06_kernel_stats_$TAG.csv 2>&1 || true
tail -3 $out/kt.log
