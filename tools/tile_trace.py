#!/usr/bin/env python3
"""Per-workgroup timelines of the gate-GEMM launches (needs a library built with -DRC_TRACE_TILES):
  cd robustcap_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DRC_TRACE_TILES -shared -o ../../tools/probe_trace.so *.cpp *.hip
  RC_LIB_PATH=$PWD/tools/probe_trace.so python tools/tile_trace.py [conf]
Prints, for the launches of one steady-state frame: workgroups, launch span, per-tile phase times (prologue / K loop /
epilogue) by tile shape, busy time per CU and the idle tail."""
import ctypes as C, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as bn
from robustcap_amd import synth, _lib
from robustcap_amd.net.sig_mp import Net

conf = sys.argv[1] if len(sys.argv) > 1 else "mixed"
B, T = 256, 48
sd, body = synth.make_state_dict(0), synth.make_body(1)
m = bn.make_inputs(body, B, T, conf, seed=2)
t = torch.from_numpy
net = Net(body=body, batch=B); net.load_state_dict(sd); net.gravityc = t(m["gravityc"])
args = [t(m[k]).cuda() for k in ("j2dc", "accc", "oric")]
ft = t(m["first_tran"]).cuda()
lib = _lib.load()
cap = 400000
buf = torch.zeros(cap * 8, dtype=torch.int64, device="cuda")
net.forward_sequence(*[a[:, :40] for a in args], first_tran=ft); torch.cuda.synchronize()
lib.rc_trace_tiles_set.argtypes = [C.c_void_p, C.c_uint64]
assert lib.rc_trace_tiles_set(C.c_void_p(buf.data_ptr()), cap) == 0
net.forward_sequence(*[a[:, 40:44] for a in args]); torch.cuda.synchronize()
rec = buf.cpu().numpy().reshape(-1, 8).astype(np.int64)
rec = rec[rec[:, 0] > 0]                                   # written slots (block id + 1)
n = len(rec)
rec = rec[np.argsort(rec[:, 4])]
TICK = 0.01   # us per wall-clock tick (100 MHz)
launches, cur, cur_end = [], [], None
for r in rec:
    if cur and r[4] > cur_end:
        launches.append(np.array(cur)); cur = []
    cur.append(r); cur_end = r[7] if len(cur) == 1 else max(cur_end, r[7])
launches.append(np.array(cur))
print("records", n, "launches", len(launches), "(4 frames x 11 expected)")
per_frame = 11
for li, L in enumerate(launches[per_frame * 2: per_frame * 3]):
    t0, t1 = L[:, 4].min(), L[:, 7].max()
    span = (t1 - t0) * TICK
    shapes = collections.defaultdict(list)
    for r in L:
        shapes[(int(r[1]) & 0xffff)].append(((r[5] - r[4]) * TICK, (r[6] - r[5]) * TICK, (r[7] - r[6]) * TICK, (r[7] - r[4]) * TICK))
    cu_busy = collections.defaultdict(float); cu_end = collections.defaultdict(int); cu_n = collections.defaultdict(int)
    for r in L:
        key = int(r[2]); cu_busy[key] += (r[7] - r[4]) * TICK; cu_end[key] = max(cu_end[key], r[7]); cu_n[key] += 1
    ends = np.array([(e - t0) * TICK for e in cu_end.values()])
    busy = np.array(list(cu_busy.values()))
    print("launch %2d: %4d WGs, span %6.1f us | CUs seen %3d, busy/CU mean %5.1f max %5.1f | last-finish per CU: mean %5.1f min %5.1f max %5.1f" % (
        li, len(L), span, len(cu_busy), busy.mean(), busy.max(), ends.mean(), ends.min(), ends.max()))
    for sh, v in sorted(shapes.items()):
        v = np.array(v)
        print("      tile %dx%-2d n=%4d  prologue %5.2f  K-loop %6.2f  epilogue %5.2f  total mean %6.2f  p10 %6.2f  p90 %6.2f us" % (
            sh >> 4, sh & 15, len(v), v[:, 0].mean(), v[:, 1].mean(), v[:, 2].mean(), v[:, 3].mean(), np.percentile(v[:, 3], 10), np.percentile(v[:, 3], 90)))
