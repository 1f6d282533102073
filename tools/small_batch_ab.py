#!/usr/bin/env python3
"""A/B of environment switches on the small-batch regime (strong-scaling shards): bench.py at a few batch sizes per setting.
    python tools/small_batch_ab.py "RC_TILE_SMALL_ROWS=1" "" -- 16 24 32 40 48 64"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
cut = args.index("--") if "--" in args else len(args)
settings, batches = args[:cut] or [""], [int(b) for b in args[cut + 1:]] or [16, 32, 64]
out = {}
for st in settings:
    env = dict(os.environ)
    for kv in st.split():
        k, v = kv.split("=", 1)
        env[k] = v
    row = {}
    for B in batches:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(B), "--steps", "256", "--warmup", "16", "--reps", "3",
                            "--no-cpu-baseline", "--no-variants"], capture_output=True, text=True, cwd=ROOT, env=env)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        row[B] = round(json.loads(line[-1])["value"] / 1e3, 1) if line else "error: " + r.stderr[-200:]
    out[st or "default"] = row
print(json.dumps(out, indent=1))
