#!/usr/bin/env python3
"""Host side of the wavefront engine at batch 256: when rc_sequence returned (everything enqueued) and when the streams were done, per tick.
    [RC_SEQ_RESIDENT=1] [RC_SEQ_H5_EARLY=1] ... python tools/host_enqueue_ab.py [conf]     (the quick A/B of this round's engine experiments,
    profiles/r06_resident_notes.txt: ~3 s per setting)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as bn
from robustcap_amd import synth
from robustcap_amd.net.sig_mp import Net
conf = sys.argv[1] if len(sys.argv) > 1 else "high"
sd, body = synth.make_state_dict(0), synth.make_body(1)
B, T = 256, 272
m = bn.make_inputs(body, B, T, conf, seed=2)
t = torch.from_numpy
net = Net(body=body, batch=B); net.load_state_dict(sd); net.gravityc = t(m["gravityc"])
a = [t(m[k]).cuda() for k in ("j2dc", "accc", "oric")]
net.forward_sequence(*[x[:, :16] for x in a], first_tran=t(m["first_tran"]).cuda()); torch.cuda.synchronize()
res = []
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    net.forward_sequence(*[x[:, 16:] for x in a])
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    res.append((t1 - t0, t2 - t0))
e, c = min(r[0] for r in res), min(r[1] for r in res)
w, s, ticks = net.sequence_stats()
print(f"resident={os.environ.get('RC_SEQ_RESIDENT','0')} {conf}: enqueue returned after {e*1e3:.2f} ms, complete after {c*1e3:.2f} ms "
      f"({(T-16)*B/c:.0f} bf/s; {c/(T-16)*1e6:.1f} us/frame, host {e/(T-16)*1e6:.1f} us/frame)")
