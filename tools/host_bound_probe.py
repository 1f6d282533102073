#!/usr/bin/env python3
"""Is sequence mode bound by the host's launch rate at small batches? Per batch: when rc_sequence returned (everything enqueued) and when
the stream was done, per frame. Round 4, MI355X: batch 1: host 35 / device 53 us per frame, 8: 35 / 66, 16: 42 / 76, 32: 51 / 89,
64: 42 / 103, 256: 90 / 228 (the host waits on full queues there) -- the device side bounds every batch.
    python tools/host_bound_probe.py"""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as bn
from robustcap_amd import synth
from robustcap_amd.net.sig_mp import Net
sd, body = synth.make_state_dict(0), synth.make_body(1)
for B in (1, 8, 16, 32, 64, 256):
    T = 272
    m = bn.make_inputs(body, B, T, "mixed", seed=2)
    t = torch.from_numpy
    net = Net(body=body, batch=B); net.load_state_dict(sd); net.gravityc = t(m["gravityc"])
    a = [t(m[k]).cuda() for k in ("j2dc", "accc", "oric")]
    net.forward_sequence(*[x[:, :16] for x in a], first_tran=t(m["first_tran"]).cuda()); torch.cuda.synchronize()
    res = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        net.forward_sequence(*[x[:, 16:] for x in a])
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        res.append((t1 - t0, t2 - t0))
    e, c = min(r[0] for r in res), min(r[1] for r in res)
    print(f"batch {B}: enqueue returned after {e*1e3:.2f} ms, complete after {c*1e3:.2f} ms ({(T-16)*B/c:.0f} bf/s; {c/(T-16)*1e6:.1f} us/frame, host {e/(T-16)*1e6:.1f} us/frame)")
    del net
