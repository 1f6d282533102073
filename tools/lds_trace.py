#!/usr/bin/env python3
"""Per-workgroup records of the shared-weight kernel while the wavefront engine runs (library built with -DRC_TRACE_TILES):
phase times by item kind, CU occupancy over a window of steady-state ticks.
  make -C robustcap_amd/csrc EXTRA=-DRC_TRACE_TILES librobustcap_hip.so && cp robustcap_amd/csrc/librobustcap_hip.so tools/probe_trace.so
  RC_LIB_PATH=$PWD/tools/probe_trace.so python tools/lds_trace.py [conf]"""
import collections
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench as bn  # noqa: E402
from robustcap_amd import _lib, synth  # noqa: E402
from robustcap_amd.net.sig_mp import Net  # noqa: E402

conf = sys.argv[1] if len(sys.argv) > 1 else "mixed"
B, T = 256, 136
sd, body = synth.make_state_dict(0), synth.make_body(1)
m = bn.make_inputs(body, B, T, conf, seed=2)
t = torch.from_numpy
net = Net(body=body, batch=B)
net.load_state_dict(sd)
net.gravityc = t(m["gravityc"])
args = [t(m[k]).cuda() for k in ("j2dc", "accc", "oric")]
lib = _lib.load()
cap = 400000
buf = torch.zeros(cap * 8, dtype=torch.int64, device="cuda")
net.forward_sequence(*[a[:, :40] for a in args], first_tran=t(m["first_tran"]).cuda())
torch.cuda.synchronize()
lib.rc_trace_lds_set.argtypes = [C.c_void_p, C.c_uint64]
assert lib.rc_trace_lds_set(C.c_void_p(buf.data_ptr()), cap) == 0
net.forward_sequence(*[a[:, 40:] for a in args])
torch.cuda.synchronize()
rec = buf.cpu().numpy().reshape(-1, 8).astype(np.int64)
rec = rec[rec[:, 0] > 0]
rec = rec[np.argsort(rec[:, 2])]
TICK = 0.01                                            # us per wall-clock tick (100 MHz)
t0, t1 = rec[:, 2].min(), rec[:, 6].max()
lo, hi = t0 + (t1 - t0) * 0.3, t0 + (t1 - t0) * 0.7
win = rec[(rec[:, 2] >= lo) & (rec[:, 6] <= hi)]
span = (hi - lo) * TICK
print(f"{len(rec)} items over {(t1 - t0) * TICK / 1e3:.2f} ms ({net.sequence_stats()}); window {span:.0f} us, {len(win)} items")
kinds = collections.defaultdict(list)
for r in win:
    H, half, ks, last = int(r[0]) & 0xffff, (int(r[0]) >> 16) & 15, (int(r[0]) >> 20) & 15, (int(r[0]) >> 24) & 1
    kinds[(H, ks, last)].append(((r[3] - r[2]) * TICK, (r[4] - r[3]) * TICK, (r[5] - r[4]) * TICK, (r[6] - r[5]) * TICK))
tot_all = 0.0
for k, v in sorted(kinds.items()):
    v = np.array(v)
    tot_all += v.sum()
    print("  H %4d ksplit %d %s n=%5d  prologue %5.2f  K loop %6.2f (p10 %6.2f p90 %6.2f)  hand-over %5.2f  epilogue %5.2f  total %6.2f us" % (
        k[0], k[1], "last " if k[2] else "first", len(v), v[:, 0].mean(), v[:, 1].mean(), np.percentile(v[:, 1], 10), np.percentile(v[:, 1], 90),
        v[:, 2].mean(), v[:, 3].mean(), v.sum(1).mean()))
busy = collections.defaultdict(float)
for r in win:
    busy[int(r[1])] += (r[6] - r[2]) * TICK
b = np.array(list(busy.values()))
print(f"  CUs seen {len(b)}: item time per CU / window mean {100 * b.mean() / span:.1f} %, min {100 * b.min() / span:.1f} %, max {100 * b.max() / span:.1f} %; items' time / (256 CUs x window) {100 * tot_all / (256 * span):.1f} %")
allv = np.array([((r[3] - r[2]), (r[4] - r[3]), (r[5] - r[4]), (r[6] - r[5])) for r in win], dtype=np.float64)
print("  all items: prologue %.1f %%, K loop %.1f %%, hand-over %.1f %%, epilogue %.1f %% of the item time" % tuple(100 * allv.sum(0) / allv.sum()))
