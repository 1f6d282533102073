#!/usr/bin/env python3
"""Per-tile records of the one-launch-per-tick kernel (library built with -DRC_TRACE_TILES): phase times by K class, the gap between
consecutive tiles of a workgroup, busy time per CU, and how a launch ends (first / last workgroup out).
  RC_LIB_PATH=$PWD/tools/probe_trace.so python tools/tick_trace.py [conf] [batch]"""
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

conf = sys.argv[1] if len(sys.argv) > 1 else "high"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
T = 136
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
lo, hi = t0 + (t1 - t0) * 0.3, t0 + (t1 - t0) * 0.7
win = rec[(rec[:, 4] >= lo) & (rec[:, 7] <= hi)]
span = (hi - lo) * TICK
print(f"{conf} batch {B}: {len(rec)} tile records over {(t1 - t0) * TICK / 1e3:.2f} ms; launch stats {net.launch_stats()}; window {span:.0f} us, {len(win)} tiles")
wide = win[(win[:, 1] & 0xffff) == 4 * 16 + 8]
pro, kl, ep = (wide[:, 5] - wide[:, 4]) * TICK, (wide[:, 6] - wide[:, 5]) * TICK, (wide[:, 7] - wide[:, 6]) * TICK
cls = np.round(kl / 8.0) * 8
for c in sorted(set(cls)):
    s = cls == c
    print(f"  64x128 tiles with a K loop of ~{c:4.0f} us: n={s.sum():6d}  prologue {pro[s].mean():5.2f}  K loop {kl[s].mean():6.2f}  reduction+epilogue {ep[s].mean():5.2f}  total {(pro + kl + ep)[s].mean():6.2f} us")
tot = pro + kl + ep
print(f"  all 64x128: prologue {pro.mean():.2f} K loop {kl.mean():.2f} reduction+epilogue {ep.mean():.2f} total {tot.mean():.2f} us -> {100 * (pro.sum() + ep.sum()) / tot.sum():.1f} % of the tiles' time outside the K loop")
# per CU: busy time and gaps between consecutive tiles
by_cu = collections.defaultdict(list)
for r in win:
    by_cu[int(r[2])].append(r)
gaps, busy = [], []
for cu, rs in by_cu.items():
    rs.sort(key=lambda r: r[4])
    busy.append(sum((r[7] - r[4]) for r in rs) * TICK)
    for a, b in zip(rs[:-1], rs[1:]):
        gaps.append((b[4] - a[7]) * TICK)
gaps, busy = np.array(gaps), np.array(busy)
print(f"  CUs seen {len(by_cu)}: tile time per CU / window mean {100 * busy.mean() / span:.1f} %, min {100 * busy.min() / span:.1f} %, max {100 * busy.max() / span:.1f} %")
small = gaps[gaps < 5.0]
print(f"  gaps between consecutive tiles of a CU: {len(gaps)} ; < 5 us: {len(small)} (median {np.median(small) if len(small) else 0:.2f} us, mean {small.mean() if len(small) else 0:.2f}); >= 5 us: {len(gaps) - len(small)} "
      f"(mean {gaps[gaps >= 5.0].mean() if (gaps >= 5).any() else 0:.1f} us, sum per CU {gaps[gaps >= 5.0].sum() / max(1, len(by_cu)):.1f} us of the {span:.0f} us window)")
# launch boundaries: cluster tile starts by the big gaps (every CU idles between two launches)
ends = np.sort(win[:, 7]); starts = np.sort(win[:, 4])
# a launch = maximal run of time with at least one tile active
ev = sorted([(r[4], 1) for r in win] + [(r[7], -1) for r in win])
act, seg_start, segs = 0, None, []
for tt, d in ev:
    if act == 0 and d == 1:
        seg_start = tt
    act += d
    if act == 0:
        segs.append((seg_start, tt))
print(f"  {len(segs)} stretches with at least one tile running; idle between them: " + ", ".join(f"{(b[0] - a[1]) * TICK:.1f}" for a, b in zip(segs[:-1], segs[1:])) + " us")
for a, b in segs[1:4]:
    inside = win[(win[:, 4] >= a) & (win[:, 7] <= b)]
    last_by_cu = collections.defaultdict(int)
    first_by_cu = {}
    for r in inside:
        cu = int(r[2]); last_by_cu[cu] = max(last_by_cu[cu], r[7]); first_by_cu[cu] = min(first_by_cu.get(cu, r[4]), r[4])
    le = np.array(list(last_by_cu.values())); fs = np.array(list(first_by_cu.values()))
    print(f"    stretch {(b - a) * TICK:.1f} us, {len(inside)} tiles on {len(last_by_cu)} CUs: first tile starts within {(fs.max() - fs.min()) * TICK:.1f} us; CUs end "
          f"{(b - np.percentile(le, 10)) * TICK:.1f} (p10) / {(b - np.percentile(le, 50)) * TICK:.1f} (p50) / {(b - np.percentile(le, 90)) * TICK:.1f} (p90) us before the last one; "
          f"tile-time / (CUs x stretch) = {100 * sum((r[7] - r[4]) for r in inside) / (len(last_by_cu) * (b - a)):.1f} %")
xcc = collections.Counter(int(r[2]) >> 8 for r in win)       # coarse: high bits of the hardware id
print(f"  hardware-id high bits (smid >> 8): {dict(sorted(xcc.items()))}")
