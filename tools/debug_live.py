#!/usr/bin/env python3
"""Debug: a reference sequence through rc_live_step with the lean capture on / off, per-frame trace and fusion state side by side."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robustcap_amd import synth
from robustcap_amd.net.sig_mp import Net
name = sys.argv[1] if len(sys.argv) > 1 else "live_pre"
s = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", f"seq_{name}.npz"))
t = torch.from_numpy
sd, body = synth.make_state_dict(0), synth.make_body(1)
nets = []
for lean in ("1", "0"):
    os.environ["RC_LIVE_LEAN"] = lean
    Net.live = str(s["live"]) == "pre"
    n = Net(body=body, batch=1); Net.live = False
    n.load_state_dict(sd)
    if str(s["live"]) == "post": n.live = True
    n.use_flat_floor = bool(s["use_flat_floor"])
    n.gravityc = t(s["gravityc"]); n.use_graph = True
    nets.append(n)
ft = t(s["first_tran"]) if s["first_tran"].size else None
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    row = []
    for n in nets:
        p, tr = n.forward_online(t(s["j2dc"][i]), t(s["accc"][i]), t(s["oric"][i]), ft if i == 0 else None, bool(s["first_frame"]) and i == 0)
        row.append((n.get_trace()[0].tolist(), n.fusion_state()[0].tolist(), n.live_stats(), p.clone(), tr.clone()))
    d = float((row[0][3] - row[1][3]).abs().max()), float((row[0][4] - row[1][4]).abs().max())
    print(i, "exp", np.round(s["trace"][i], 3).tolist(), "| lean", row[0][0], row[0][1], row[0][2], "| full", row[1][0], row[1][1], "| d", d)
