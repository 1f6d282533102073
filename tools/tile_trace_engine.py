#!/usr/bin/env python3
"""Per-workgroup records of the gate-GEMM tiles while the wavefront engine runs (library built with -DRC_TRACE_TILES, see
tools/tile_trace.py): phase times by tile shape and K, busy time per CU over a window of steady-state ticks.
  RC_LIB_PATH=$PWD/tools/probe_trace.so python tools/tile_trace_engine.py [conf]"""
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
lib.rc_trace_tiles_set.argtypes = [C.c_void_p, C.c_uint64]
assert lib.rc_trace_tiles_set(C.c_void_p(buf.data_ptr()), cap) == 0
net.forward_sequence(*[a[:, 40:] for a in args])
torch.cuda.synchronize()
rec = buf.cpu().numpy().reshape(-1, 8).astype(np.int64)
rec = rec[rec[:, 0] > 0]
rec = rec[np.argsort(rec[:, 4])]
TICK = 0.01                                            # us per wall-clock tick (100 MHz)
t0, t1 = rec[:, 4].min(), rec[:, 7].max()
lo, hi = t0 + (t1 - t0) * 0.3, t0 + (t1 - t0) * 0.7     # steady state: the middle of the call
win = rec[(rec[:, 4] >= lo) & (rec[:, 7] <= hi)]
span = (hi - lo) * TICK
print(f"{len(rec)} tile records over {(t1 - t0) * TICK / 1e3:.2f} ms ({net.sequence_stats()}); window {span:.0f} us, {len(win)} tiles")
shapes = collections.defaultdict(list)
for r in win:
    shapes[(int(r[1]) & 0xffff, round(float((r[6] - r[5]) * TICK), -1))].append(((r[5] - r[4]) * TICK, (r[6] - r[5]) * TICK, (r[7] - r[6]) * TICK))
agg = collections.defaultdict(list)
for (sh, _), v in shapes.items():
    agg[sh] += v
for sh, v in sorted(agg.items()):
    v = np.array(v)
    tot = v.sum(1)
    print("  tile %dx%-2d n=%5d  prologue %5.2f  K loop %6.2f (p10 %6.2f p90 %6.2f)  reduction+epilogue %5.2f  total %6.2f us  -> %4.1f %% of the tiles' time outside the K loop" % (
        sh >> 4, sh & 15, len(v), v[:, 0].mean(), v[:, 1].mean(), np.percentile(v[:, 1], 10), np.percentile(v[:, 1], 90), v[:, 2].mean(), tot.mean(),
        100 * (v[:, 0].sum() + v[:, 2].sum()) / tot.sum()))
busy = collections.defaultdict(float)
for r in win:
    busy[int(r[2])] += (r[7] - r[4]) * TICK
b = np.array(list(busy.values()))
print(f"  CUs seen {len(b)}: wide-tile time per CU / window mean {100 * b.mean() / span:.1f} %, min {100 * b.min() / span:.1f} %, max {100 * b.max() / span:.1f} %")
allv = np.array([((r[5] - r[4]) * TICK, (r[6] - r[5]) * TICK, (r[7] - r[6]) * TICK) for r in win])
print(f"  all tiles: prologue {100 * allv[:, 0].sum() / allv.sum():.1f} %, K loop {100 * allv[:, 1].sum() / allv.sum():.1f} %, reduction + epilogue {100 * allv[:, 2].sum() / allv.sum():.1f} % of the tile time")
