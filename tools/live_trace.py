#!/usr/bin/env python3
"""Where the latency-bound kernels of the lean live frame spend their time: in-kernel wall-clock stamps (100 MHz) of block 0 /
thread 0, from a -DRC_LIVE_TRACE build of the library (tools/probe_livetrace.so; RC_LIB_PATH selects it):
    hipcc ... -DRC_LIVE_TRACE -shared -o tools/probe_livetrace.so <csrc sources> -lhsa-runtime64
    RC_LIB_PATH=tools/probe_livetrace.so python tools/live_trace.py [conf=high] [period_ms=0]
period_ms > 0: the frames arrive every period_ms (16.667 = 60 fps: the device idles in between and rc_live_step runs the pre-step).
RC_TRACE_SYNC=1: rc_live_step through ctypes with host stamps around it, placed on the device's clock (tools/clock_sync): how long from the
call to K1's first instruction, and from K7's last stamp to the return."""
import time
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from robustcap_amd import synth  # noqa: E402
from robustcap_amd.net.sig_mp import Net  # noqa: E402

NAMES = {0: ["entry", "inputs arrived + prep", "LDS rows ready (barrier)", "end"],
         1: ["entry", "partials arrived, slices in LDS (barrier)", "sums done (barrier)", "input rows built (barrier)", "end"],
         2: ["entry", "partials + body constants arrived (barrier)", "sub-net outputs in LDS (barrier)", "row words handed out",
             "6D -> R (sync)", "IK + foot chains", "translation / floor logic", "pose stored", "mesh landmarks (if needed)", "end"],
         3: ["l0 entry", "l0 first A loads issued", "l0 K loop done", "l0 reduction barrier", "l0 end",
             "l1 entry", "l1 first A loads issued", "l1 K loop done", "l1 reduction barrier", "l1 end"]}
KNAME = {0: "K1 rc_live_k1 (prep + linear1)", 1: "K4 rc_live_k4 (sums + fuse + linear1)", 2: "K7 rc_live_k7 (sums + tail)",
         3: "LSTM tile 0 of the second stage (rnn6, the last launches to stamp)"}


def clock_offset():
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "clock_sync", "libclock_sync.so"))
    out = (C.c_double * 3)()
    rc = lib.clock_sync(out)
    assert rc == 0, rc
    return out[0], out[1], int(out[2])


def main_sync(conf, period):
    """Host stamps around rc_live_step (C ABI, preallocated tensors) against the device stamps of the same frame."""
    sd, body = synth.make_state_dict(0), synth.make_body(1)
    m = synth.make_motion(7, 1, 300, body, conf=conf)
    net = Net(body=body, batch=1)
    net.load_state_dict(sd)
    net.gravityc = torch.from_numpy(m["gravityc"])
    net.use_graph = True
    t = torch.from_numpy
    T = m["j2dc"].shape[1]
    ins = [(t(m["j2dc"][0, k]).contiguous(), t(m["accc"][0, k]).contiguous(), t(m["oric"][0, k]).contiguous()) for k in range(T)]
    pose, tran = torch.empty(1, 24, 3, 3), torch.empty(1, 3)
    net.forward_online(*ins[0], first_frame=True)
    fn, ctx = net._lib.rc_live_step, net._ctx
    rd = C.CDLL(os.environ["RC_LIB_PATH"]).rc_live_trace_read
    pp, pt = C.c_void_p(pose.data_ptr()), C.c_void_p(tran.data_ptr())
    ptrs = [(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(c.data_ptr())) for a, b, c in ins]
    off0, nspt, moved = clock_offset()
    ts0 = time.perf_counter_ns()
    rows = []
    buf = (C.c_ulonglong * 64)()
    t_next = time.perf_counter() + max(period, 1e-3)
    for i in range(1, 260):
        a, b, c = ptrs[i]
        slack = t_next - time.perf_counter() - 1e-3
        if period > 0 and slack > 0:
            time.sleep(slack)
        if period > 0:
            while time.perf_counter() < t_next:
                pass
            t_next += period
        t0 = time.perf_counter_ns()
        rc = fn(ctx, a, b, c, None, 0, pp, pt)
        t1 = time.perf_counter_ns()
        assert rc == 0
        assert rd(buf) == 0
        st = np.array(buf[:], dtype=np.float64).reshape(4, 16)
        if i >= 60:
            rows.append((t0, t1, st[0, 0] * nspt, st[2, 9] * nspt, st[0, 3] * nspt, st[0, 5] * nspt))
    off1, _, _ = clock_offset()
    ts1 = time.perf_counter_ns() - 50e6                                    # (the second calibration takes 50 ms; its best sample can be anywhere in it)
    r = np.array(rows)
    if len(r) == 0:
        print('no frames'); return
    lean, full = net.live_stats()
    # the two clocks drift (~10 ppm: tens of us over a paced run): the offset of a frame is interpolated between the two calibrations
    off = off0 + (off1 - off0) * (r[:, 0] - ts0) / max(ts1 - ts0, 1.0)
    sub = (r[:, 2] + off - r[:, 0]) * 1e-3
    gpu = (r[:, 3] - r[:, 2]) * 1e-3
    fin = (r[:, 1] - (r[:, 3] + off)) * 1e-3
    ok = (gpu > 0) & (gpu < 1e6)                                           # (frames off the lean plan leave stale stamps)
    print(f"clock offset {off0 * 1e-3:.2f} us before, {off1 * 1e-3:.2f} us after the run ({moved} samples each; interpolated per frame; it contains one posted write over PCIe "
          f"and, paced, up to ~0.5 us of interpolation error: 'call -> K1' reads that much long and 'K7 -> return' that much short)")
    print(f"  call -> K1 entry      p50 {np.percentile(sub[ok], 50):6.2f}  mean {sub[ok].mean():6.2f} us")
    print(f"  K1 entry -> K7 end    p50 {np.percentile(gpu[ok], 50):6.2f}  mean {gpu[ok].mean():6.2f} us")
    k1e = (r[:, 4] + off - r[:, 0]) * 1e-3
    rest = (r[:, 3] - r[:, 4]) * 1e-3
    print(f"  call -> K1 end        p50 {np.percentile(k1e[ok], 50):6.2f} us      K1 end -> K7 end  p50 {np.percentile(rest[ok], 50):6.2f} us")
    if os.environ.get("RC_LIVE_SPIN"):
        seen = (r[:, 5] + off - r[:, 0]) * 1e-3
        print(f"  call -> K1 has the command (block 0)  p50 {np.percentile(seen[ok], 50):6.2f} us")
    print(f"  K7 end -> return      p50 {np.percentile(fin[ok], 50):6.2f}  mean {fin[ok].mean():6.2f} us")
    print(f"  call -> return        p50 {np.percentile((r[:, 1] - r[:, 0])[ok] * 1e-3, 50):6.2f} us   ({int(ok.sum())} frames, lean/full {lean}/{full}, pre-steps {net.live_prestep_stats()[0]})")


def main():
    conf = sys.argv[1] if len(sys.argv) > 1 else "high"
    period = float(sys.argv[2]) * 1e-3 if len(sys.argv) > 2 else 0.0
    if os.environ.get("RC_TRACE_SYNC"):
        return main_sync(conf, period)
    sd, body = synth.make_state_dict(0), synth.make_body(1)
    m = synth.make_motion(7, 1, 200, body, conf=conf)
    net = Net(body=body, batch=1)
    net.load_state_dict(sd)
    net.gravityc = torch.from_numpy(m["gravityc"])
    net.use_graph = True
    t = torch.from_numpy
    acc = np.zeros((4, 16))
    n = 0
    for i in range(200):
        if period > 0:
            time.sleep(period)
        net.forward_online(t(m["j2dc"][0, i]), t(m["accc"][0, i]), t(m["oric"][0, i]), first_frame=(i == 0))
        if i < 60:
            continue
        buf = (C.c_ulonglong * 64)()
        assert net._lib._handle if False else True
        rc = C.CDLL(os.environ["RC_LIB_PATH"]).rc_live_trace_read(buf)
        assert rc == 0
        a = np.array(buf[:], dtype=np.float64).reshape(4, 16)
        for k in range(4):
            nn = len(NAMES[k])
            acc[k, :nn] += (a[k, :nn] - a[k, 0]) * 0.01          # us from the kernel's first stamp
        n += 1
    print(f"lean live frame, conf={conf}, {n} frames{', one every %.2f ms' % (period * 1e3) if period > 0 else ' back to back'}, pre-steps {net.live_prestep_stats()[0]}, "
          f"us from each kernel's entry stamp (block 0, thread 0):")
    for k in range(4):
        print(KNAME[k])
        prev = 0.0
        for j, name in enumerate(NAMES[k]):
            v = acc[k, j] / n
            print(f"   {v:7.2f}  (+{v - prev:5.2f})  {name}")
            prev = v
    print("lean/full frames:", net.live_stats())


if __name__ == "__main__":
    main()
