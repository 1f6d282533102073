#!/bin/bash
# What would a time-hoisted engine gain? (x . W_ih of every layer step as tall GEMMs over chunks of frames, only h . W_hh per frame.)
# Upper bound: the engine as it is with the input half of every layer step LEFT OUT (a -DRC_SKIP_XHALF build: wrong results, right timing) =
# the per-frame cost of a hoisted engine whose tall GEMMs, pre-activation round trips and chunk pipeline fill were free.
#   hipcc ... -DRC_SKIP_XHALF -shared -o tools/probe_skipx.so <sources>   (robustcap_amd/csrc/Makefile's command line)
#   bash tools/hoist_bound.sh
for conf in high mixed; do
  for lib in "" tools/probe_skipx.so; do
    echo "$conf ${lib:-product}: $(RC_LIB_PATH=${lib:+$PWD/$lib} python bench.py --steps 512 --warmup 16 --reps 3 --conf $conf --no-cpu-baseline --no-variants 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(round(d["value"]), "bf/s", d["ms_per_step"]*1e3, "us/frame")')"
  done
done
