#!/usr/bin/env python3
"""A/B of library builds and environment switches on one box: bench.py per setting, settings interleaved over the rounds.
    python tools/ab.py [--steps 512] [--rounds 2] [--conf mixed] [--batch 256] "label|lib.so|ENV=1 ENV2=x" ...
(lib empty = the in-tree library). Prints one line per run and the per-setting medians."""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
opt = {"--steps": "512", "--rounds": "2", "--conf": "mixed", "--batch": "256", "--reps": "3", "--warmup": "16"}
while args and args[0] in opt:
    opt[args[0]] = args[1]
    args = args[2:]
settings = []
for a in args:
    label, lib, envs = (a.split("|") + ["", ""])[:3]
    settings.append((label, lib, dict(kv.split("=", 1) for kv in envs.split())))
res = {s[0]: [] for s in settings}
for rnd in range(int(opt["--rounds"])):
    for label, lib, envs in settings:
        env = dict(os.environ, **envs)
        if lib:
            env["RC_LIB_PATH"] = os.path.join(ROOT, lib)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", opt["--batch"], "--steps", opt["--steps"], "--warmup", opt["--warmup"],
                            "--reps", opt["--reps"], "--conf", opt["--conf"], "--no-cpu-baseline", "--no-variants"], capture_output=True, text=True, cwd=ROOT, env=env)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if not line:
            print(f"{label}: ERROR {r.stderr[-400:]}", flush=True)
            continue
        d = json.loads(line[-1])
        rf = d.get("roofline") or {}
        res[label].append(d["value"])
        print(f"{label:28s} {d['value'] / 1e3:9.1f} k bf/s  ms/step {d['ms_per_step']:.4f}  kernel {rf.get('kernel')}  launches/step {rf.get('launches_per_step')}  "
              f"avg_launch_us {rf.get('avg_launch_us')}  frac {rf.get('frac')}  union {rf.get('union', {}).get('frac') if isinstance(rf.get('union'), dict) else None}  "
              f"ticks {rf.get('engine', {}).get('ticks') if isinstance(rf.get('engine'), dict) else None}", flush=True)
print(json.dumps({k: {"median": statistics.median(v), "runs": v} for k, v in res.items() if v}))
