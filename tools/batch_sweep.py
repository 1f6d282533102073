#!/usr/bin/env python3
"""Frame rate of the sig_mp path against the batch: bench.py workloads at several batch sizes in one process tree.
    python tools/batch_sweep.py [conf] > profiles/rNN_batch_sweep.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    conf = sys.argv[1] if len(sys.argv) > 1 else "mixed"
    rows = []
    for B in (1, 8, 16, 32, 64, 128, 256, 512, 1024, 2048):
        steps = 96 if B >= 1024 else 256
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(B), "--steps", str(steps), "--warmup", "16",
                            "--conf", conf, "--no-cpu-baseline", "--no-variants"], capture_output=True, text=True, cwd=ROOT)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            rows.append({"batch": B, "error": r.stderr[-300:]})
            continue
        d = json.loads(line[-1])
        rf = d["roofline"] or {}
        rows.append({"batch": B, "body_frames_per_s": d["value"], "ms_per_step": d["ms_per_step"], "gemm_frac": rf.get("frac"),
                     "path_frac": rf.get("path_frac"), "products": "split-bf16" if "split" in (rf.get("kernel") or "") else "fp32 MFMA"})
    print(json.dumps({"command": f"python bench.py --batch B --steps 256 (96 for B >= 1024) --warmup 16 --conf {conf} --no-cpu-baseline --no-variants",
                      "note": "contexts of batch >= 48 use the split-bf16 products, smaller ones the fp32 MFMA (launch-chain and weight-streaming bound)",
                      "sweep": rows}, indent=1))


if __name__ == "__main__":
    main()
