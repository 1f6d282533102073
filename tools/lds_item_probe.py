#!/usr/bin/env python3
"""One sub-net step at a time through rc_lstm_step at batch 256: under `rocprofv3 --kernel-trace` the duration of a shared-weight
launch that holds ONE layer step -- i.e. of one work item (prologue + K loop + hand-over + epilogue) with the chip to itself.
    rocprofv3 --kernel-trace --stats -d out -o kt -- python tools/lds_item_probe.py ; python tools/lds_item_probe.py --read out/**/kt_results.db"""
import glob
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--read":
    db = sqlite3.connect(sys.argv[2]); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    scol = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
    name_col = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else "name")
    rows = cur.execute(f"select s.{name_col}, d.start, d.end, d.grid_size_x, d.workgroup_size_x from {disp} d join {sym} s on d.kernel_id = s.id order by d.start").fetchall()
    agg = {}
    for n, a, b, g, w in rows:
        if "rc_gemm" in n:
            agg.setdefault((n.split("(")[0][:40], g // max(w, 1)), []).append((b - a) / 1e3)
    for (n, wg), v in sorted(agg.items()):
        v = v[len(v) // 2:]                                # second half: warm
        print(f"{n:42s} {wg:5d} wg  x{len(v):3d}  avg {sum(v) / len(v):8.2f} us  min {min(v):8.2f}  max {max(v):8.2f}")
    sys.exit(0)

import torch
from robustcap_amd import synth
from robustcap_amd.net.sig_mp import Net

B = int(os.environ.get("PROBE_B", "256"))
net = Net(body=synth.make_body(1), batch=B)
net.load_state_dict(synth.make_state_dict(0))
net.set_gemm_mode(True)
dims = {"rnn2": 72, "rnn3": 141, "rnn4": 171, "rnn6": 240, "rnn7": 141, "rnn8": 141}
for name, k in dims.items():
    x = torch.randn(B, k, device="cuda")
    for _ in range(12):
        net.lstm_step(name, x)
torch.cuda.synchronize()
print("done")
