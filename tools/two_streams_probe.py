#!/usr/bin/env python3
"""Experiment: one context of B bodies vs S independent contexts of B/S bodies on S streams (S host threads)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as bn
from robustcap_amd import synth
from robustcap_amd.net.sig_mp import Net

B, T = 256, 272
conf = sys.argv[1] if len(sys.argv) > 1 else "mixed"
sd, body = synth.make_state_dict(0), synth.make_body(1)
m = bn.make_inputs(body, B, T, conf, seed=2)
t = torch.from_numpy
dev = torch.device("cuda")
full = [t(m[k]).to(dev) for k in ("j2dc", "accc", "oric")]
ft = t(m["first_tran"]).to(dev)
grav = t(m["gravityc"])

def make(rows):
    net = Net(body=body, batch=rows.stop - rows.start)
    net.load_state_dict(sd)
    net.gravityc = grav[rows]
    return net

for S in (1, 2, 4):
    parts = [slice(i * B // S, (i + 1) * B // S) for i in range(S)]
    nets = [make(r) for r in parts]
    streams = [torch.cuda.Stream() for _ in parts]
    ins = [[x[r].contiguous() for x in full] for r in parts]
    def work(i, lo, hi, first):
        with torch.cuda.stream(streams[i]):
            nets[i].forward_sequence(ins[i][0][:, lo:hi], ins[i][1][:, lo:hi], ins[i][2][:, lo:hi], first_tran=ft[parts[i]] if first else None)
    def run_all(lo, hi, first):
        th = [threading.Thread(target=work, args=(i, lo, hi, first)) for i in range(S)]
        [x.start() for x in th]; [x.join() for x in th]
        torch.cuda.synchronize()
    run_all(0, 16, True)
    t0 = time.perf_counter()
    run_all(16, T, False)
    dt = time.perf_counter() - t0
    print("S=%d contexts of %d bodies: %.1f us per 256-body frame, %.0f body-frames/s" % (S, B // S, dt / (T - 16) * 1e6, B * (T - 16) / dt))
