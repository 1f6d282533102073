#!/usr/bin/env python3
"""Where the latency-bound kernels of the lean live frame spend their time: in-kernel wall-clock stamps (100 MHz) of block 0 /
thread 0, from a -DRC_LIVE_TRACE build of the library (tools/probe_livetrace.so; RC_LIB_PATH selects it):
    hipcc ... -DRC_LIVE_TRACE -shared -o tools/probe_livetrace.so <csrc sources> -lhsa-runtime64
    RC_LIB_PATH=tools/probe_livetrace.so python tools/live_trace.py [conf=high] [period_ms=0]
period_ms > 0: the frames arrive every period_ms (16.667 = 60 fps: the device idles in between and rc_live_step runs the pre-step)."""
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


def main():
    conf = sys.argv[1] if len(sys.argv) > 1 else "high"
    period = float(sys.argv[2]) * 1e-3 if len(sys.argv) > 2 else 0.0
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
