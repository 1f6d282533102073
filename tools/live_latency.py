#!/usr/bin/env python3
"""BASELINE config 5: batch 1, 60 fps streaming, per-frame latency p50/p99 of forward_online (host tensors in,
host tensors out). Prints one JSON line:
  graph_c_abi   rc_live_step alone through ctypes on preallocated host tensors (what a C caller of the ABI sees; the default
                configuration: steady-state frames on the lean seven-launch capture, the rest on the frame-stepped captures)
  graph         the same through Net.forward_online (Python surface)
  eager         Net.forward_online without use_graph (rc_step)
  variants      C-ABI p50/p99 under environment switches (RC_LIVE_LEAN=0: round 3's plan, RC_LIVE_LEAN_NC=2, RC_LIVE_AQL=0: hipGraph replay,
                RC_LIVE_DONE_FLAG=0: completion by the queue's signal instead of the word the last kernel stores, RC_LIVE_EAGER=1)
    python tools/live_latency.py [frames=10000] [conf=mixed] [variants=1]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from robustcap_amd import synth  # noqa: E402
from robustcap_amd.net.sig_mp import Net  # noqa: E402


def run(net, m, n):
    t = torch.from_numpy
    T = m["j2dc"].shape[1]
    lat = np.empty(n)
    for i in range(n):
        k = i % T
        args = (t(m["j2dc"][0, k]), t(m["accc"][0, k]), t(m["oric"][0, k]))
        t0 = time.perf_counter()
        net.forward_online(*args, first_frame=(i == 0))
        lat[i] = time.perf_counter() - t0
    return lat[50:] * 1e6


def run_c(net, m, n, period_s=0.0):
    """The library call alone (rc_live_step through ctypes on preallocated host tensors): what a C caller of the ABI sees.
    period_s > 0: the frames ARRIVE every period_s (60 fps: 1 / 60) -- the caller sleeps until shortly before the arrival, spins to it,
    and the latency is measured from the arrival to the outputs on the host, like live_server.py:40-48 receiving a camera frame."""
    import ctypes as C
    t = torch.from_numpy
    T = m["j2dc"].shape[1]
    ins = [(t(m["j2dc"][0, k]).contiguous(), t(m["accc"][0, k]).contiguous(), t(m["oric"][0, k]).contiguous()) for k in range(T)]
    pose, tran = torch.empty(1, 24, 3, 3), torch.empty(1, 3)
    net.forward_online(*ins[0], first_frame=True)                       # captures the frame
    fn, ctx = net._lib.rc_live_step, net._ctx
    pp, pt = C.c_void_p(pose.data_ptr()), C.c_void_p(tran.data_ptr())
    ptrs = [(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(c.data_ptr())) for a, b, c in ins]
    lat = np.empty(n)
    t_next = time.perf_counter() + period_s
    for i in range(n):
        a, b, c = ptrs[1 + i % (T - 1)]
        if period_s > 0:
            slack = t_next - time.perf_counter() - 1e-3
            if slack > 0:
                time.sleep(slack)
            while time.perf_counter() < t_next:
                pass
            t0 = t_next
            t_next += period_s
        else:
            t0 = time.perf_counter()
        rc = fn(ctx, a, b, c, None, 0, pp, pt)
        lat[i] = time.perf_counter() - t0
        assert rc == 0
    return lat[50:] * 1e6


def stats(lat):
    return {"p50_us": round(float(np.percentile(lat, 50)), 1), "p99_us": round(float(np.percentile(lat, 99)), 1),
            "mean_us": round(float(lat.mean()), 1), "fps": round(1e6 / float(lat.mean()), 1)}


def make(sd, body, m, graph=True, env=None):
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        net = Net(body=body, batch=1)                                   # the switches are read when the context is created
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    net.load_state_dict(sd)
    net.gravityc = torch.from_numpy(m["gravityc"])
    net.use_graph = graph
    return net


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    conf = sys.argv[2] if len(sys.argv) > 2 else "mixed"
    variants = (int(sys.argv[3]) if len(sys.argv) > 3 else 1) != 0
    sd, body = synth.make_state_dict(0), synth.make_body(1)
    m = synth.make_motion(7, 1, 600, body, conf=conf)
    out = {}
    for mode in ("graph", "graph_c_abi", "eager"):
        net = make(sd, body, m, graph=(mode != "eager"))
        lat = run_c(net, m, n) if mode == "graph_c_abi" else run(net, m, n if mode != "eager" else min(n, 3000))
        out[mode] = stats(lat)
        if mode == "graph_c_abi":
            lean, full = net.live_stats()
            out[mode]["lean_frames"], out[mode]["full_frames"] = lean, full
            import ctypes as C
            prof = (C.c_double * 4)()
            net._lib.rc_get_live_profile(net._ctx, prof)
            out[mode]["host_us"] = {"stage": round(prof[0], 2), "enqueue": round(prof[1], 2), "wait": round(prof[2], 2), "copy_out": round(prof[3], 2)}
            a, b, note = C.c_int32(0), C.c_int32(0), C.create_string_buffer(256)
            net._lib.rc_get_live_backend(net._ctx, C.byref(a), C.byref(b), note, 256)
            out[mode]["backend"] = {"lean_captured": a.value, "aql": b.value, "note": note.value.decode()}
        del net
    # BASELINE config 5 as stated: 60 fps. The frames arrive every 16.67 ms; between two frames the device idles (clocks, caches).
    n_paced = int(os.environ.get("RC_PACED_FRAMES", "1200"))
    if n_paced > 0:
        for name, env in (("paced_60fps", {}), ("paced_60fps_graph_replay", {"RC_LIVE_AQL": "0"}), ("paced_60fps_no_prestep", {"RC_LIVE_PRESTEP": "0"}), ("paced_60fps_prestep_warm", {"RC_LIVE_PREWARM": "1"})):
            net = make(sd, body, m, env=env)
            out[name] = stats(run_c(net, m, n_paced, 1.0 / 60.0))
            out[name]["frames"] = n_paced - 50
            del net
    if variants:
        out["variants"] = {}
        for name, env in (("lean_off", {"RC_LIVE_LEAN": "0"}), ("lean_nc2", {"RC_LIVE_LEAN_NC": "2"}), ("lean_graph", {"RC_LIVE_AQL": "0"}), ("lean_signal", {"RC_LIVE_DONE_FLAG": "0"}), ("lean_edge_agent", {"RC_AQL_EDGE_SCOPE": "agent"}),
                          ("lean_direct_launches", {"RC_LIVE_EAGER": "1"}), ("lean_again", {})):
            net = make(sd, body, m, env=env)
            out["variants"][name] = stats(run_c(net, m, min(n, 4000)))
            del net
    print(json.dumps({"metric": "forward_online latency, batch 1, host->host", "frames": n, "conf": conf, "weights_MB": 243.06,
                      "weight_stream_floor_us_at_6.3TBps": 38.6, **out}))


if __name__ == "__main__":
    main()
