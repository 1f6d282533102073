#!/bin/bash
# A/B of environment switches on the bench's long variants: tools/ab_env.sh "NAME=VAL ..." "NAME=VAL ..." ...   (one bench run per argument)
for cfg in "$@"; do
  env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline > /tmp/ab.json 2>/dev/null
  python - "$cfg" <<'PY'
import json, sys
d = json.loads(open('/tmp/ab.json').read().strip().split('\n')[-1])
v = d['variants']
print(f"{sys.argv[1]:60s} k20 {d['value']/1e3:7.1f}k  mixed_long {v['mixed_long']['value']/1e3:7.1f}k  high_long {v['high_long']['value']/1e3:7.1f}k  occ1024 {v['occ1024']['value']/1e3:7.1f}k  launch {d['roofline']['avg_launch_us']:.1f} us")
PY
done
