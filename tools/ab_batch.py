#!/usr/bin/env python3
"""One batch size, one confidence schedule, 256-frame calls on the wavefront engine: body-frames/s (best of 4). The quick A/B behind the
thresholds of the shared-weight kernel (RC_LDS_MIN_BATCH / RC_LDS_MIN_ROWS / RC_LDS_KSPLIT_*; rc_api.cpp, profiles/r06_batch_sweep.json):
    [RC_LDS_MIN_BATCH=..] [RC_LDS_KSPLIT_1024=1] python tools/ab_batch.py <batch> <mixed|high|occ>"""
import sys, os, time, torch
sys.path.insert(0, '/root/repo')
import bench as bn
from robustcap_amd import synth
from robustcap_amd.net.sig_mp import Net
B = int(sys.argv[1]); conf = sys.argv[2]
sd, body = synth.make_state_dict(0), synth.make_body(1)
T = 272
m = bn.make_inputs(body, B, T, conf, seed=2)
t = torch.from_numpy
net = Net(body=body, batch=B); net.load_state_dict(sd); net.gravityc = t(m["gravityc"])
a = [t(m[k]).cuda() for k in ("j2dc", "accc", "oric")]
net.forward_sequence(*[x[:, :16] for x in a], first_tran=t(m["first_tran"]).cuda()); torch.cuda.synchronize()
best = 1e9
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    net.forward_sequence(*[x[:, 16:] for x in a]); torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
print(f"B {B} {conf} minrows {os.environ.get('RC_LDS_MIN_ROWS','160')}: {(T-16)*B/best:.0f} bf/s")
