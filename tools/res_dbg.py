#!/usr/bin/env python3
"""Debug helper for the resident engine: batch 256, wavefront engine vs frame-stepped launches on one schedule, largest differences.
    RC_SEQ_RESIDENT=1 python tools/res_dbg.py [frames] [conf]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as bn
from robustcap_amd import synth
from robustcap_amd.net.sig_mp import Net
sd, body = synth.make_state_dict(0), synth.make_body(1)
B, T = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 24
conf = sys.argv[2] if len(sys.argv) > 2 else "high"
m = bn.make_inputs(body, B, T, conf, seed=2)
t = torch.from_numpy
outs = []
for seq in (True, False):
    net = Net(body=body, batch=B); net.load_state_dict(sd); net.gravityc = t(m["gravityc"])
    net.set_sequence_mode(seq, 8, force=True)
    a = [t(m[k]).cuda() for k in ("j2dc", "accc", "oric")]
    p, tr = net.forward_sequence(*a, first_tran=t(m["first_tran"]).cuda()); torch.cuda.synchronize()
    st = [net.get_state(n) for n in ("rnn2", "rnn3", "rnn4", "rnn6", "rnn7", "rnn8")]
    outs.append((p.cpu(), tr.cpu(), st, net.sequence_stats()))
(p1, t1, s1, st1), (p2, t2, s2, st2) = outs
print("stats", st1, st2)
d = (p1 - p2).abs().amax(dim=(0, 2, 3, 4)) if p1.dim() == 5 else (p1 - p2).abs().flatten(2).amax(dim=(0, 2))
print("pose max diff per frame:", [f"{float(x):.1e}" for x in d])
print("tran max diff:", float((t1 - t2).abs().max()))
for n, a, b in zip(("rnn2", "rnn3", "rnn4", "rnn6", "rnn7", "rnn8"), s1, s2):
    print(n, "h diff", float((a[0] - b[0]).abs().max()), "c diff", float((a[1] - b[1]).abs().max()))
