#!/usr/bin/env python3
"""batch-1 sequence (512 frames, one enqueue): per-kernel profile target for the live path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robustcap_amd import synth
from robustcap_amd.net.sig_mp import Net
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
conf = sys.argv[2] if len(sys.argv) > 2 else "high"
sd, body = synth.make_state_dict(0), synth.make_body(1)
m = synth.make_motion(7, B, 512, body, conf=conf)
t = torch.from_numpy
net = Net(body=body, batch=B); net.load_state_dict(sd); net.gravityc = t(m["gravityc"])
args = [t(m[k]).cuda() for k in ("j2dc", "accc", "oric")]
for rep in range(3):
    net.reset_states(); torch.cuda.synchronize(); t0 = time.perf_counter()
    net.forward_sequence(*args, first_frame=True); torch.cuda.synchronize()
    print("B=%d %s: %.1f us/frame" % (B, conf, (time.perf_counter() - t0) / 512 * 1e6))
