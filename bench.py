#!/usr/bin/env python3
"""Benchmark of the sig_mp per-frame path on MI355X: body-frames/s at batch 256 (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong] [--conf mixed|high|occ]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one frame of the hot path (prep -> 6 LSTM sub-nets -> fusion/FK tail -> vision updater) for one batch
of bodies, inputs resident in HBM. Workload = BASELINE.json configs[1]: synthetic 60 fps sequences, 6 IMUs + 33
keypoints, batch 256 x 512 frames, mixed-confidence schedule (50 % high / 20 % mid /
30 % occluded, which forces the frame-stepped path incl. the vision updater); the all-visible variant (config 2a,
which the sequence-mode engine accelerates) is timed too and reported under "variants".

Scaling: ``weak`` (default) = 256 bodies on EVERY rank; ``strong`` = 256 bodies in total, split over the ranks with
dist.shard_range. For N > 1 the outputs are gathered to rank 0 (RCCL) inside the timed region, in four asynchronous
chunks that overlap the remaining frames.

Rank 0 prints ONE JSON line (metric, value, ..., roofline, cpu_baseline). The K-step call is timed ``--reps`` times
(default 5; every repetition = reset + W warm-up frames + K timed frames between barrier + synchronize) and ``value`` /
``ms_per_step`` are those of the MEDIAN repetition (``timing`` carries min / max). ``variants`` carries the other
configurations of BASELINE.json on the same box: ``high`` (all-visible, same K), ``fp32_mfma``, ``mixed_long`` /
``high_long`` (512 frames per call whatever --steps is: configs[1]'s own T, the wavefront engine's steady state), ``occ1024`` (config 4) and
``live_b1`` (config 5: p50 / p99 of the captured batch-1 frame; ``graph_replay`` = the same capture through hipGraphLaunch). The roofline pass and the side legs are guarded: a
failure there is recorded as {"error": ...} and the line is still printed.
"""
import argparse
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from robustcap_amd import config as C  # noqa: E402
from robustcap_amd import dist as rdist  # noqa: E402
from robustcap_amd import synth  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: f32-input MFMA (16x16x4 / 32x32x2), dense
PEAK_BF16_MFMA_TFLOPS = 2500.0         # dense bf16 MFMA; the split kernel issues 6 bf16 MFMAs per fp32-equivalent product
LONG_FRAMES = 512                      # frames per call of the *_long variants: the T of BASELINE configs[1]
PRODUCTS_SPLIT = ("fp32 operands split exactly into 3 bf16 terms each; every product = 6 partial products on "
                  "v_mfma_f32_16x16x32_bf16 (all terms >= 2^-16 of the product), fp32 accumulation, fp32 gates / state")
PRODUCTS_FP32 = "v_mfma_f32_16x16x4_f32 (fp32 operands, an fma chain per output element)"
PACED_FRAMES = 1000                    # live_b1.paced_60fps: frames at 60 fps (16.7 s of the default run)
CPU_FRAMES_BATCHED = 8                 # cpu_baseline: 5 samples of B x 8 frames batched (median; round-4 review: three samples spread by
CPU_FRAMES_SINGLE = 48                 # 20 %) + 48 frames batch-1 (~20-30 s of CPU work)
CPU_SAMPLES = 5


def pmc_traffic(batch, conf, steps):
    """Fabric-side bytes per gate-GEMM launch from a committed rocprofv3 PMC pass of THIS workload (batch, confidence
    schedule AND frames per call must match the keys stored with the measurement: the launch population -- launches per
    step, FLOPs per launch -- changes with the call length), else None: PMC counters cannot be read from inside this
    process, and a figure measured on another workload is not evidence for this run."""
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
        if not (name.endswith(".json") and "pmc_traffic" in name):
            continue
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            if int(d.get("batch", -1)) == batch and d.get("conf") == conf and int(d.get("steps", -1)) == steps:
                return float(d["traffic_bytes_per_launch"]), name
        except (OSError, KeyError, ValueError, TypeError):
            continue
    return None, None


def make_inputs(body, B, T, conf, seed, unique=32):
    """[B,T,...] synthetic inputs. `unique` distinct motions are generated and tiled over the batch with
    per-body confidence schedules (generation cost is host-side numpy, outside any timed region)."""
    u = min(B, unique)
    m = synth.make_motion(seed, u, T, body, conf=conf)
    rep = (B + u - 1) // u
    out = {k: np.concatenate([v] * rep, 0)[:B].copy() for k, v in m.items()}
    if B > u:   # decorrelate the regimes of the tiled copies: per-body confidence schedule
        for b in range(u, B):
            c = synth.conf_schedule(seed * 7919 + b, 7, T, conf)
            out["j2dc"][b, :, :, 2] = np.clip(out["j2dc"][b, :, :, 2] - out["conf"][b][:, None] + c[:, None], 0, 1)
        out["conf"] = out["j2dc"][..., 2].mean(-1)
    return out


def make_inputs_device(body, B, T, conf, seed, unique=32):
    """``make_inputs`` with forward kinematics, IMU synthesis and projection on the GPU (robustcap_amd.preprocess.make_motion_device: SURVEY.md
    8(f) rank 3); the inputs are born in HBM. Same seeds, same trajectories, same tiling of `unique` motions with per-body confidence
    schedules; equal to the host generator to fp32 rounding (tests/test_gpu_evaluate.py). Returns device tensors j2dc / accc / oric and
    host arrays gravityc / first_tran."""
    from robustcap_amd import preprocess
    u = min(B, unique)
    m = preprocess.make_motion_device(seed, u, T, body, conf=conf)
    rep = (B + u - 1) // u
    out = {k: (torch.cat([v] * rep, 0)[:B].clone() if torch.is_tensor(v) else np.concatenate([v] * rep, 0)[:B].copy()) for k, v in m.items()}
    if B > u:   # decorrelate the regimes of the tiled copies: per-body confidence schedule
        c_new = np.stack([synth.conf_schedule(seed * 7919 + b, 7, T, conf) for b in range(u, B)]).astype(np.float32)           # [B-u, T]
        shift = torch.from_numpy(c_new - out["conf"][u:]).to(out["j2dc"].device)
        out["j2dc"][u:, :, :, 2] = torch.clamp(out["j2dc"][u:, :, :, 2] + shift[:, :, None], 0.0, 1.0)
        out["conf"] = out["j2dc"][..., 2].mean(-1).cpu().numpy()
    return out


def cpu_baseline(sd, body, m, frames_batched=CPU_FRAMES_BATCHED, frames_single=CPU_FRAMES_SINGLE, samples=CPU_SAMPLES):
    """The oracle (a port, parity-pinned to the reference) on this host's cores: batched (median of ``samples``
    back-to-back samples of ``frames_batched`` frames each) and batch-1 frame by frame like evaluate.py.
    ``m`` only has to hold ONE frame more than a sample; shorter inputs shorten the sample."""
    from oracle import sig_mp_oracle as O
    t = torch.from_numpy
    B, T = m["j2dc"].shape[:2]
    if T < 2:
        raise ValueError("cpu_baseline needs at least 2 frames of input")
    frames_batched, frames_single = min(frames_batched, T - 1), min(frames_single, T - 1)
    threads = torch.get_num_threads()
    net = O.OracleNet(body, batch=B)
    net.load_numpy_state_dict(sd)
    net.gravityc = t(m["gravityc"])
    net.forward_batch(t(m["j2dc"][:, 0]), t(m["accc"][:, 0]), t(m["oric"][:, 0]), None, True)
    rates, secs = [], []
    for s_ in range(max(1, samples)):
        t0 = time.perf_counter()
        for q in range(frames_batched):
            i = 1 + (s_ * frames_batched + q) % (T - 1)
            net.forward_batch(t(m["j2dc"][:, i]), t(m["accc"][:, i]), t(m["oric"][:, i]))
        dt_b = time.perf_counter() - t0
        rates.append(B * frames_batched / dt_b)
        secs.append(dt_b)
    one = O.OracleNet(body, batch=1)
    one.load_numpy_state_dict(sd)
    one.forward_online(t(m["j2dc"][0, 0]), t(m["accc"][0, 0]), t(m["oric"][0, 0]), None, True)
    t0 = time.perf_counter()
    for i in range(1, 1 + frames_single):
        one.forward_online(t(m["j2dc"][0, i]), t(m["accc"][0, i]), t(m["oric"][0, i]))
    dt_1 = time.perf_counter() - t0
    # the same on ONE thread ({1 thread, all threads}); a handful of frames, the thread count restored afterwards
    n_1t = max(1, min(8, frames_single))
    torch.set_num_threads(1)
    try:
        t0 = time.perf_counter()
        for i in range(1, 1 + n_1t):
            one.forward_online(t(m["j2dc"][0, i]), t(m["accc"][0, i]), t(m["oric"][0, i]))
        dt_1t = time.perf_counter() - t0
    finally:
        torch.set_num_threads(threads)
    return {"value": round(float(np.median(rates)), 1), "unit": "body-frames/s", "cores": threads, "kind": "port",
            "samples": len(rates), "min": round(min(rates), 1), "max": round(max(rates), 1),
            "batch1_all_threads": round(frames_single / dt_1, 1), "batch1_one_thread": round(n_1t / dt_1t, 1),
            "sample": f"oracle (torch CPU, oneDNN LSTM) batched B={B} x {frames_batched} frames, median of {len(rates)} samples "
                      f"({', '.join('%.1fs' % x for x in secs)}); "
                      f"batch-1 frame-by-frame like evaluate.py: {frames_single / dt_1:.1f} body-frames/s "
                      f"({frames_single} frames = {dt_1:.1f}s), on one thread {n_1t / dt_1t:.1f} ({n_1t} frames = {dt_1t:.1f}s); nproc={os.cpu_count()}"}


def guarded(fn, *a, **k):
    """fn(*a, **k), or {"error": ...} -- the bench line must be printed whatever the side legs do."""
    try:
        return fn(*a, **k)
    except Exception as e:  # noqa: BLE001 - deliberately broad: report, do not die
        traceback.print_exc(file=sys.stderr)
        return {"error": f"{type(e).__name__}: {e}"}


class Workload:
    """One confidence schedule: inputs resident on the device, a context of this rank's rows, timed/instrumented runs."""

    def __init__(self, sd, body, conf, B, W, frames, rank, world, seed_base=2, split=None):
        from robustcap_amd.net.sig_mp import Net
        self.conf, self.B, self.W, self.rank, self.world = conf, B, W, rank, world
        dev = torch.device("cuda")
        self.dev = dev
        if os.environ.get("RC_BENCH_HOST_INPUTS"):                          # (A/B: the host generator)
            self.m = make_inputs(body, B, W + frames, conf, seed=seed_base + rank)
            self.j2d, self.acc, self.ori = (torch.from_numpy(self.m[k]).to(dev) for k in ("j2dc", "accc", "oric"))
        else:
            self.m = make_inputs_device(body, B, W + frames, conf, seed=seed_base + rank)
            self.j2d, self.acc, self.ori = (self.m[k] for k in ("j2dc", "accc", "oric"))
        self.ft = torch.from_numpy(self.m["first_tran"]).to(dev)
        self.net = Net(body=body, batch=B)
        self.net.load_state_dict(sd)
        if split is not None:
            self.net.set_gemm_mode(split)
        self.split = bool(self.net.gemm_mode)
        self.net.gravityc = torch.from_numpy(self.m["gravityc"])

    def run(self, lo, hi, first):
        return self.net.forward_sequence(self.j2d[:, lo:hi], self.acc[:, lo:hi], self.ori[:, lo:hi],
                                         first_tran=self.ft if first else None)

    def sync(self):
        if self.world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed(self, K, gather_rows_total=None):
        """W untimed warmup frames, then exactly K timed frames between barrier + synchronize; max over ranks."""
        B, W, world, rank = self.B, self.W, self.world, self.rank
        self.net.reset_states()
        if W > 0:
            self.run(0, W, True)
        self.sync()
        t0 = time.perf_counter()
        if world > 1:
            # The path's only exchange: the outputs go to rank 0 (RCCL over xGMI). The K steps are enqueued in up to 4 chunks
            # of at least 16 frames (a chunk is one planned rc_sequence call: shorter ones would fall back to the
            # frame-stepped launches) and each chunk's gather starts as soon as its kernels are queued, so all but the last
            # transfer hide behind compute. Row blocks may differ by one row under strong scaling: gather_rows pads.
            n_chunks = max(1, min(4, K // 16))
            edges = [W + (K * c) // n_chunks for c in range(n_chunks + 1)]
            parts = []
            for ci, (lo, hi) in enumerate(zip(edges[:-1], edges[1:])):
                if hi > lo:
                    p, tr = self.run(lo, hi, W == 0 and ci == 0)
                    if gather_rows_total is None:
                        parts.append((rdist.RowGather(p.reshape(B, -1)), rdist.RowGather(tr.reshape(B, -1))))
                    else:
                        parts.append((p, tr))
            if gather_rows_total is None:
                outs = [(gp.result(), gt.result()) for gp, gt in parts]
            else:
                outs = [(rdist.gather_rows(p.reshape(B, -1), gather_rows_total, dst=0),
                         rdist.gather_rows(tr.reshape(B, -1), gather_rows_total, dst=0)) for p, tr in parts]
            pose, tran = (outs[-1] if rank == 0 else (p, tr))
        else:
            pose, tran = self.run(W, W + K, W == 0)
        self.sync()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], device=self.dev if torch.distributed.get_backend() == "nccl" else "cpu", dtype=torch.float64)
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
            dt = float(tmax.item())
        assert torch.isfinite(pose).all() and torch.isfinite(tran).all()
        return dt

    def timed_reps(self, K, reps, gather_rows_total=None):
        """`reps` repetitions of timed(K); returns the sorted list of their times."""
        return sorted(self.timed(K, gather_rows_total) for _ in range(max(1, reps)))

    def roofline(self, K, dt, bodies_total):
        """Dominant kernel: HIP-event timing (on the launch stream, inside the library) of every wide-tile gate-GEMM
        launch over the same K steps. rc_gemm_kernel runs linear1 and both LSTM layers of all six sub-nets: 99.7 % of
        the algorithmic FLOPs. The 16-row launches (transition steps, linear2) run on rc_gemm_small_kernel, a
        weight-streaming kernel outside this roofline; `path_frac` is the whole frame (every kernel, the timed
        region's own clock) against the same peak."""
        B, W, net = self.B, self.W, self.net
        net.reset_states()
        if W > 0:
            self.run(0, W, True)
        torch.cuda.synchronize()
        lds = net.launch_stats()[0] > 0          # this context's LSTM layer steps run on the shared-weight kernel: that is the dominant kernel
        net.gemm_timing(3 if lds else 2)
        self.run(W, W + K, W == 0)
        torch.cuda.synchronize()
        ms, launches = net.gemm_timing_read()
        busy_ms = net.gemm_timing_busy()
        net.gemm_timing(0)
        if launches <= 0 or ms <= 0 or busy_ms <= 0:
            raise RuntimeError("no gate-GEMM launch was timed")
        flop_per_launch = B * (C.FLOPS_LSTM_PER_BODY_FRAME if lds else C.FLOPS_PER_BODY_FRAME - C.FLOPS_LINEAR2_PER_BODY_FRAME) * K / launches
        avg_s = ms * 1e-3 / launches
        # The wavefront engine issues the wide launches of a tick on two / three streams: they share the chip, so a launch's own
        # duration covers time in which the other one holds part of the CUs. The kernel's rate is its FLOPs over the time during
        # which it runs at all (union of the launch intervals); the per-launch figure is kept beside it (it is what a rocprofv3
        # kernel summary shows: avg_launch_us there = avg_launch_us here).
        ach_union = flop_per_launch * launches / (busy_ms * 1e-3) / 1e12
        ach = flop_per_launch / avg_s / 1e12
        path = bodies_total * K * C.FLOPS_PER_BODY_FRAME / dt / 1e12 / self.world
        traffic, src = pmc_traffic(B, self.conf, K)
        issued_peak = PEAK_BF16_MFMA_TFLOPS / 6.0 if self.split else PEAK_FP32_MFMA_TFLOPS
        wave, stepped, ticks = net.sequence_stats()
        return {"bound": "mfma", "kernel": net.gemm_kernel_name(), "achieved": round(ach, 2),
                "peak": round(issued_peak, 1),
                "unit": "TFLOP/s", "frac": round(ach / issued_peak, 4), "frac_per_launch": round(ach / issued_peak, 4),
                "peak_fp32_input": PEAK_FP32_MFMA_TFLOPS, "frac_fp32_roof": round(ach_union / PEAK_FP32_MFMA_TFLOPS, 4),
                "traffic": traffic,
                "traffic_note": (f"fabric-side L2 miss bytes per gate-GEMM launch, rocprofv3 PMC pass of this batch/schedule "
                                 f"(profiles/{src})" if src else
                                 "no PMC pass committed for this batch / schedule / frames per call (profiles/*pmc_traffic*.json are keyed by all three)"),
                "avg_launch_us": round(avg_s * 1e6, 2), "launches": launches, "launches_per_step": round(launches / K, 2),
                "busy_ms": round(busy_ms, 3), "concurrency": round(ms / busy_ms, 3),
                "union": {"achieved": round(ach_union, 2), "frac": round(ach_union / issued_peak, 4)},
                "flop_per_launch": flop_per_launch,
                "path_achieved": round(path, 2), "path_frac": round(path / issued_peak, 4),
                "products": PRODUCTS_SPLIT if self.split else PRODUCTS_FP32,
                "engine": {"wavefront_frames": wave, "frame_stepped_frames": stepped, "ticks": ticks},
                "note": "achieved: the algorithmic fp32 FLOPs of one wide-tile gate-GEMM launch / the average HIP-event duration of such a "
                        "launch (avg_launch_us: what a rocprofv3 --kernel-trace --stats summary of the same command shows for the kernel); "
                        "peak: the roof of the instructions the kernel issues (split mode: dense bf16 MFMA peak / 6 partial products = "
                        "416.7 TFLOP/s fp32-equivalent; fp32 mode: the 157.3 TFLOP/s fp32-input MFMA peak); frac = achieved / peak PER "
                        "LAUNCH. Since round 6 a tick is THREE launches of this kernel on three streams (rnn4 | rnn6 | the H = 512 nets) "
                        "that run side by side, each on its share of the CUs: a launch's own duration is `concurrency` (= sum of the "
                        "launch durations / the time at least one runs) times what its work takes on the whole chip, so the kernel's MFMA "
                        "utilisation ON THE CHIP is union.frac = the same FLOPs over the time during which at least one such launch runs "
                        "(round 5: two launches per tick, per-launch 0.21, union 0.33); frac_fp32_roof: "
                        "the union rate against the fp32-input roof (rounds 1-4 quoted it as frac; it may exceed 1 in split mode); "
                        "path_frac: whole frame incl. the weight-streaming 16-row launches and the per-frame logic kernels, against peak"}


def self_launch(n):
    """`python bench.py --gpus N ...` without a launcher (WORLD_SIZE unset): one process per GPU via torch.distributed.run on a
    free local port -- the same command line the driver uses for N > 1. Fails loudly when fewer than N devices are visible
    (RC_DIST_SHARE_DEVICE=1, dry runs on a 1-GPU box, lets the ranks share device 0)."""
    import socket
    have = torch.cuda.device_count()
    if have < n and os.environ.get("RC_DIST_SHARE_DEVICE") != "1":
        raise SystemExit(f"bench.py: --gpus {n} but only {have} GPU(s) are visible to this process")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execvpe(sys.executable, cmd, env)


def rate(dts, K, bodies_total, what):
    """variant record from the sorted times of its repetitions (median = the value)."""
    med = dts[len(dts) // 2]
    return {"value": round(bodies_total * K / med, 1), "ms_per_step": round(med / K * 1e3, 4), "frames": K, "reps": len(dts),
            "min": round(bodies_total * K / dts[-1], 1), "max": round(bodies_total * K / dts[0], 1), "workload": what}


def live_b1(sd, body, frames=2000):
    """BASELINE config 5: batch 1, one captured frame per host round trip through the C ABI (rc_live_step on host tensors, like
    live_server.py:40-48 hands them over): p50 / p99 latency of `frames` frames. The steady-state frame is the lean seven-launch
    capture (csrc/rc_live.hip), dispatched as a pre-built AQL packet chain (csrc/rc_aql.cpp); `graph_replay` = the same capture
    replayed with hipGraphLaunch (RC_LIVE_AQL=0), fewer frames."""
    import ctypes as C_
    from robustcap_amd.net.sig_mp import Net
    m = synth.make_motion(7, 1, 600, body, conf="mixed")
    t = torch.from_numpy
    T = m["j2dc"].shape[1]
    ins = [(t(m["j2dc"][0, k]).contiguous(), t(m["accc"][0, k]).contiguous(), t(m["oric"][0, k]).contiguous()) for k in range(T)]

    def run(n_frames, env, period_s=0.0):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            net = Net(body=body, batch=1)                                   # the switches are read when the context is created
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        net.load_state_dict(sd)
        net.gravityc = t(m["gravityc"])
        net.use_graph = True
        pose, tran = torch.empty(1, 24, 3, 3), torch.empty(1, 3)
        net.forward_online(*ins[0], first_frame=True)                       # captures the frame
        fn, ctx = net._lib.rc_live_step, net._ctx
        pp, pt = C_.c_void_p(pose.data_ptr()), C_.c_void_p(tran.data_ptr())
        ptrs = [(C_.c_void_p(a.data_ptr()), C_.c_void_p(b.data_ptr()), C_.c_void_p(c.data_ptr())) for a, b, c in ins]
        lat = np.empty(n_frames + 50)
        prof = np.zeros((n_frames + 50, 6))
        last = (C_.c_double * 6)()
        get_last = getattr(net._lib, "rc_get_live_last_profile", None) if period_s > 0 else None
        t_next = time.perf_counter() + period_s
        for i in range(n_frames + 50):
            a, b, c = ptrs[1 + i % (T - 1)]
            if period_s > 0:                                                # the frame ARRIVES at t_next: sleep, spin to it, clock from the arrival
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
            if rc != 0:
                raise RuntimeError(f"rc_live_step failed ({rc})")
            if get_last is not None:                                        # (outside the clocked span)
                get_last(ctx, last)
                prof[i] = last[:]
        lat = lat[50:] * 1e6
        lean, full = net.live_stats()
        cap, aql, note = C_.c_int32(0), C_.c_int32(0), C_.create_string_buffer(256)
        net._lib.rc_get_live_backend(net._ctx, C_.byref(cap), C_.byref(aql), note, 256)
        extra = {"prof": prof[50:], "spin": net.live_spin_stats(), "replayed": net.live_replayed(), "presteps": net.live_prestep_stats()[0]}
        return lat, lean, full, bool(cap.value), bool(aql.value), note.value.decode(), extra

    def paced_record(env, what, frames=None):
        """config 5 as BASELINE states it: a frame every 16.67 ms, the device idle in between; latency from the frame's ARRIVAL. Beside the
        percentiles: where the slowest 1 % of the frames spent their time on the host (rc_get_live_last_profile) -- a slow frame whose
        `wait` segment (enqueue done -> completion seen; the device time is inside it) is as short as everybody's was late BEFORE it
        reached the library (the pacing loop's own wake-up), one whose wait is long was slow on the queue / device."""
        lat, lean, full, _, _, _, ex = run(frames or PACED_FRAMES, env, 1.0 / 60.0)
        pr = ex["prof"]
        host_in = lat - pr[:, :4].sum(1)                                    # arrival -> rc_live_step entered + return -> clock read: the caller's side
        slow = lat >= np.percentile(lat, 99)
        def seg(sel):                                                       # (the AQL path submits and polls in one call: its time is `frame_us`)
            a = pr[sel, :4].mean(0)
            return {"stage_us": round(float(a[0]), 1), "frame_us": round(float(a[1] + a[2]), 1), "copy_out_us": round(float(a[3]), 1)}
        edges = [0, 60, 70, 80, 90, 100, 125, 150, 200, 400, 1e9]
        hist = np.histogram(lat, bins=edges)[0]
        taken, lost = ex["spin"]
        return {"p50_us": round(float(np.percentile(lat, 50)), 1), "p99_us": round(float(np.percentile(lat, 99)), 1),
                "p99_9_us": round(float(np.percentile(lat, 99.9)), 1), "max_us": round(float(lat.max()), 1),
                "mean_us": round(float(lat.mean()), 1), "frames": len(lat), "period_ms": round(1e3 / 60.0, 3),
                "histogram_us": {f"<{int(b)}" if b < 1e9 else f">={int(a)}": int(n) for a, b, n in zip(edges[:-1], edges[1:], hist)},
                "lean_frames": lean, "full_frames": full, "spin_taken": taken, "spin_lost": lost, "replayed": ex["replayed"], "presteps": ex["presteps"],
                "all_frames": dict(seg(slice(None)), outside_library_us=round(float(host_in.mean()), 1)),
                "slowest_1pct": dict(seg(slow), outside_library_us=round(float(host_in[slow].mean()), 1), frames=int(slow.sum()),
                                     library_max_us=round(float(pr[slow, :4].sum(1).max()), 1)),
                "mode": what,
                "note": "inputs arrive every 16.67 ms (sleep + spin to the arrival time), latency = arrival -> outputs on the host. Host split of "
                        "rc_live_step (rc_get_live_last_profile): stage_us = inputs staged + capture chosen, frame_us = packets pushed -> completion "
                        "seen by the polling host (the frame's device time is inside it), copy_out_us; outside_library_us = the part of the latency "
                        "spent before rc_live_step was entered and after it returned (the pacing loop's own wake-up); library_max_us = the longest "
                        "time any of the slowest frames spent inside the library"}

    lat, lean, full, cap, aql, note, _ = run(frames, {})
    out = {"p50_us": round(float(np.percentile(lat, 50)), 1), "p99_us": round(float(np.percentile(lat, 99)), 1),
           "mean_us": round(float(lat.mean()), 1), "frames": frames, "value": round(1e6 / float(lat.mean()), 1),
           "unit": "body-frames/s", "weight_stream_floor_us": 38.6,
           "lean_frames": lean, "full_frames": full, "launches_per_lean_frame": 7 if cap else None,
           "dispatch": "AQL packet chain on the context's own HSA queue" if aql else ("hipGraphLaunch" + (f" ({note})" if note else "")),
           "workload": "BASELINE config 5: batch 1, captured frame (steady state: seven kernels), host tensors in / out through rc_live_step"}
    out["paced_60fps"] = paced_record({}, "default: idle-time pre-step + armed queue; no kernel left waiting on the device between frames")
    if aql:
        out["paced_60fps_spin"] = paced_record({"RC_LIVE_SPIN": "1"}, "RC_LIVE_SPIN=1 (opt-in): the next frame is queued ahead, its first kernel polls a "
                                               "mailbox on the device until the inputs arrive (~84 workgroups busy between frames)",
                                               frames=max(250, PACED_FRAMES * 2 // 5))       # (the opt-in leg: 6.7 s of the default run instead of 16.7)
    if aql:
        lat2 = run(max(200, frames // 4), {"RC_LIVE_AQL": "0"})[0]
        out["graph_replay"] = {"p50_us": round(float(np.percentile(lat2, 50)), 1), "p99_us": round(float(np.percentile(lat2, 99)), 1),
                               "frames": len(lat2)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the timed K-step call (value = median)")
    ap.add_argument("--batch", type=int, default=256, help="bodies per GPU (weak) or in total (strong)")
    ap.add_argument("--conf", default="mixed", choices=["mixed", "high", "occ"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--gemm-mode", default="auto", choices=["auto", "split", "fp32"],
                    help="product arithmetic of the GEMMs (auto: split-bf16 products from 48 bodies in total)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the variants (other schedules / configs)")
    args = ap.parse_args()
    if args.steps < 1 or args.warmup < 0 or args.batch < 1 or args.reps < 1:
        ap.error("--steps >= 1, --warmup >= 0, --batch >= 1, --reps >= 1")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)              # does not return: re-executes this command line under torch.distributed.run
    rank, world, local = rdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher's WORLD_SIZE is {world}")
    torch.cuda.set_device(local if world > 1 else 0)
    ranks_seen = None
    if world > 1:                           # every rank is really there: a sum of ones over the job's own collective backend
        one = torch.ones(1, dtype=torch.int32, device="cuda" if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(one)
        ranks_seen = int(one.item())

    K, W = args.steps, args.warmup
    if args.scaling == "strong":
        a, b = rdist.shard_range(args.batch, rank, world)
        B, bodies_total = b - a, args.batch
        if B < 1:
            raise SystemExit(f"strong scaling: {args.batch} bodies cannot be split over {world} ranks")
    else:
        B, bodies_total = args.batch, args.batch * world
    sd, body = synth.make_state_dict(0), synth.make_body(1)
    # equal row blocks use the asynchronous RowGather; uneven strong splits (batch % world != 0) the padded gather
    strong_total = bodies_total if (args.scaling == "strong" and world > 1 and args.batch % world) else None
    # The product arithmetic follows the TOTAL workload, not this rank's shard: the same rows give the same bits on 1 or N GPUs.
    from robustcap_amd.net.sig_mp import Net
    split_total = Net.default_gemm_mode(bodies_total) if args.gemm_mode == "auto" else args.gemm_mode == "split"
    long_frames = LONG_FRAMES if (not args.no_variants and K < LONG_FRAMES) else 0

    main_w = Workload(sd, body, args.conf, B, W, max(K, long_frames), rank, world, split=split_total)
    dts = main_w.timed_reps(K, args.reps, strong_total)
    dt = dts[len(dts) // 2]
    roof = guarded(main_w.roofline, K, dt, bodies_total) if rank == 0 else None

    variants = None
    if not args.no_variants:
        v = {}

        def variant(w, k, what, reps=3):
            d = guarded(w.timed_reps, k, reps, strong_total)
            return d if isinstance(d, dict) else rate(d, k, bodies_total, what)
        if long_frames:
            v[f"{args.conf}_long"] = variant(main_w, long_frames, f"the main schedule at {long_frames} frames per call (the wavefront "
                                             "engine's steady state; the driver's --steps is shorter than its fill + drain)")
        if main_w.split:
            main_w.net.set_gemm_mode(False)
            v["fp32_mfma"] = variant(main_w, K, "the main workload with the products on the fp32-input MFMA instead of the "
                                                "split-bf16 partial products (rc_set_gemm_mode 0): bitwise fma chains")
            main_w.net.set_gemm_mode(True)
        del main_w.j2d, main_w.acc, main_w.ori
        if args.conf != "high":
            hw = guarded(Workload, sd, body, "high", B, W, max(K, long_frames), rank, world, 2, split_total)
            if isinstance(hw, dict):
                v["high"] = hw
            else:
                v["high"] = variant(hw, K, "SURVEY.md 8(d) config 2a: every frame visible (c >= 0.8), same batch and frame count")
                if long_frames:
                    v["high_long"] = variant(hw, long_frames, f"config 2a at {long_frames} frames per call")
                del hw
        if world > 1 and args.scaling == "weak":
            # the same bodies-in-total as ONE GPU's batch, split over the ranks (dist.shard_range): what strong scaling of the
            # headline workload gives; the weak figure above keeps --batch bodies on every rank
            def strong():
                if args.batch < world:                                    # (the same verdict on every rank: no rank may skip a collective)
                    raise RuntimeError(f"{args.batch} bodies cannot be split over {world} ranks")
                a, b = rdist.shard_range(args.batch, rank, world)
                sw = Workload(sd, body, args.conf, b - a, W, max(K, long_frames), rank, world, 3, Net.default_gemm_mode(args.batch))
                tot = args.batch if args.batch % world else None
                r = {"k": rate(sw.timed_reps(K, 3, tot), K, args.batch, f"strong scaling: {args.batch} bodies in total over {world} ranks, "
                                                                       f"{K} frames per call")}
                if long_frames:
                    r["long"] = rate(sw.timed_reps(long_frames, 3, tot), long_frames, args.batch, f"the same at {long_frames} frames per call")
                return r
            sv = guarded(strong)
            if "k" in sv:
                v["strong"] = sv["k"]
                if "long" in sv:
                    v["strong_long"] = sv["long"]
            else:
                v["strong"] = sv
        if world == 1:
            def occ():
                w = Workload(sd, body, "occ", 1024, 8, 64, 0, 1)
                return rate(w.timed_reps(64, 3), 64, 1024, "BASELINE config 4: batch 1024, occlusion-masked keypoints (runs of "
                            "c <= 0.7: the confidence-gated branch + vision updater), 64 frames per call")
            v["occ1024"] = guarded(occ)
            v["live_b1"] = guarded(live_b1, sd, body)
        variants = v if rank == 0 else None

    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            m_cpu = make_inputs(body, B, 1 + max(CPU_SAMPLES * CPU_FRAMES_BATCHED, CPU_FRAMES_SINGLE), args.conf, seed=2)
            cpu = guarded(cpu_baseline, sd, body, m_cpu)
        value = bodies_total * K / dt
        print(json.dumps({
            "metric": "body-frames/sec (sig_mp fwd + FK) at batch 256", "value": round(value, 1), "unit": "body-frames/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "timing": {"reps": len(dts), "stat": "median", "call_ms": round(dt * 1e3, 4), "min_call_ms": round(dts[0] * 1e3, 4),
                       "max_call_ms": round(dts[-1] * 1e3, 4)},
            "products": PRODUCTS_SPLIT if main_w.split else PRODUCTS_FP32,
            "config": {"workload": f"synthetic 60 fps, 6 IMU + 33 keypoints, batch {B} x {K} frames per GPU "
                                   f"({bodies_total} bodies in total, {args.scaling} scaling), "
                                   f"confidence schedule '{args.conf}', seeded random weights (63.4 M params)",
                       "batch_per_gpu": B, "bodies_total": bodies_total, "frames": K, "conf": args.conf,
                       "parallelism": f"dp{world} (sequence sharding)"},
            "rccl": None if world == 1 else {"backend": torch.distributed.get_backend(), "ranks_seen": ranks_seen},
            "roofline": roof, "cpu_baseline": cpu, "variants": variants}), flush=True)
    if world > 1:
        torch.distributed.barrier()          # rank 0 may still be in its instrumented pass
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
