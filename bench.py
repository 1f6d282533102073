#!/usr/bin/env python3
"""Benchmark of the sig_mp per-frame path on MI355X: body-frames/s at batch 256 (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one frame of the hot path (prep -> 6 LSTM sub-nets -> fusion/FK tail -> vision updater) for one batch
of 256 bodies per GPU, inputs resident in HBM. Workload = BASELINE.json configs[1]: synthetic 60 fps sequences,
6 IMUs + 33 keypoints, batch 256 x 512 frames, mixed-confidence schedule (50 % high /
20 % mid / 30 % occluded, which forces the frame-stepped path incl. the vision updater). Weak scaling: every rank
runs its own 256 bodies; for N > 1 the outputs are gathered to rank 0 (RCCL) inside the timed region, in four
asynchronous chunks that overlap the remaining frames.

Rank 0 prints ONE JSON line (metric, value, ..., roofline, cpu_baseline).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from robustcap_amd import config as C  # noqa: E402
from robustcap_amd import dist as rdist  # noqa: E402
from robustcap_amd import synth  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: f32-input MFMA (16x16x4 / 32x32x2), dense


def pmc_traffic():
    """Fabric-side bytes per gate-GEMM launch from the committed rocprofv3 PMC passes (FETCH_SIZE doubled per the gfx950
    correction + WRITE_SIZE, x1024), or None. PMC counters cannot be read from inside this process."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            return float(json.load(f)["traffic_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


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


def cpu_baseline(sd, body, m, frames_batched=16, frames_single=96):
    """The oracle (a port, parity-pinned to the reference) on this host's cores: batched B=256 and batch-1."""
    from oracle import sig_mp_oracle as O
    t = torch.from_numpy
    B = m["j2dc"].shape[0]
    threads = torch.get_num_threads()
    net = O.OracleNet(body, batch=B)
    net.load_numpy_state_dict(sd)
    net.gravityc = t(m["gravityc"])
    net.forward_batch(t(m["j2dc"][:, 0]), t(m["accc"][:, 0]), t(m["oric"][:, 0]), None, True)
    t0 = time.perf_counter()
    for i in range(1, 1 + frames_batched):
        net.forward_batch(t(m["j2dc"][:, i]), t(m["accc"][:, i]), t(m["oric"][:, i]))
    dt_b = time.perf_counter() - t0
    one = O.OracleNet(body, batch=1)
    one.load_numpy_state_dict(sd)
    one.forward_online(t(m["j2dc"][0, 0]), t(m["accc"][0, 0]), t(m["oric"][0, 0]), None, True)
    t0 = time.perf_counter()
    for i in range(1, 1 + frames_single):
        one.forward_online(t(m["j2dc"][0, i]), t(m["accc"][0, i]), t(m["oric"][0, i]))
    dt_1 = time.perf_counter() - t0
    return {"value": round(B * frames_batched / dt_b, 1), "unit": "body-frames/s", "cores": threads, "kind": "port",
            "sample": f"oracle (torch CPU, oneDNN LSTM) batched B={B} x {frames_batched} frames = {dt_b:.1f}s; "
                      f"batch-1 frame-by-frame like evaluate.py: {frames_single / dt_1:.1f} body-frames/s "
                      f"({frames_single} frames = {dt_1:.1f}s); nproc={os.cpu_count()}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--conf", default="mixed", choices=["mixed", "high", "occ"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank, world, local = rdist.init_from_env()
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    torch.cuda.set_device(local if world > 1 else 0)
    from robustcap_amd.net.sig_mp import Net

    B, K, W = args.batch, args.steps, args.warmup
    T = W + K
    sd, body = synth.make_state_dict(0), synth.make_body(1)
    m = make_inputs(body, B, T, args.conf, seed=2 + rank)
    dev = torch.device("cuda")
    j2d, acc, ori = (torch.from_numpy(m[k]).to(dev) for k in ("j2dc", "accc", "oric"))
    ft = torch.from_numpy(m["first_tran"]).to(dev)
    net = Net(body=body, batch=B)
    net.load_state_dict(sd)
    net.gravityc = torch.from_numpy(m["gravityc"])

    def run(lo, hi, first):
        return net.forward_sequence(j2d[:, lo:hi], acc[:, lo:hi], ori[:, lo:hi], first_tran=ft if first else None)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # ---- warmup (the first W frames of the sequences), then exactly K timed steps -------------------------
    run(0, W, True)
    sync()
    t0 = time.perf_counter()
    if world > 1:
        # The path's only exchange: the outputs go to rank 0 (RCCL over xGMI). The K steps are enqueued in 4 chunks and
        # each chunk's gather starts as soon as its kernels are queued, so all but the last transfer hide behind compute.
        edges = [W + (K * c) // 4 for c in range(5)]
        gathers = []
        for lo, hi in zip(edges[:-1], edges[1:]):
            if hi > lo:
                p, tr = run(lo, hi, False)
                gathers.append((rdist.RowGather(p.reshape(B, -1)), rdist.RowGather(tr.reshape(B, -1)), hi - lo))
        parts = [(gp.result(), gt.result(), n) for gp, gt, n in gathers]
        if rank == 0:
            pose = torch.cat([p.view(world * B, n, 24, 3, 3) for p, _, n in parts], dim=1)
            tran = torch.cat([q.view(world * B, n, 3) for _, q, n in parts], dim=1)
        else:
            pose, tran = p, tr
    else:
        pose, tran = run(W, T, False)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev if torch.distributed.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    assert torch.isfinite(pose).all() and torch.isfinite(tran).all()

    # ---- dominant kernel: HIP-event timing of every rc_gemm_kernel launch over the same K steps -----------
    # rc_gemm_kernel (wide MFMA tiles) runs linear1 and both LSTM layers of all six sub-nets: 99.7 % of the algorithmic
    # FLOPs in 6 launches per frame. The 16-row launches (the three transition launches, the two linear2 launches) run
    # on rc_gemm_small_kernel, a weight-streaming kernel, and are not part of this roofline; `path_frac` is the whole
    # frame (every kernel, the timed region's own clock) against the same peak.
    roof = None
    if rank == 0:
        net.reset_states()
        run(0, W, True)
        torch.cuda.synchronize()
        net.gemm_timing(2)
        run(W, T, False)
        torch.cuda.synchronize()
        ms, launches = net.gemm_timing_read()
        net.gemm_timing(0)
        flop_per_launch = B * (C.FLOPS_PER_BODY_FRAME - C.FLOPS_LINEAR2_PER_BODY_FRAME) * K / launches
        avg_s = ms * 1e-3 / launches
        ach = flop_per_launch / avg_s / 1e12
        path = world * B * K * C.FLOPS_PER_BODY_FRAME / dt / 1e12 / world
        roof = {"bound": "mfma", "kernel": "rc_gemm_kernel", "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": pmc_traffic(),
                "traffic_note": "fabric-side L2 miss bytes per gate-GEMM launch (both kernels), rocprofv3 PMC "
                                "(profiles/r01_pmc_traffic.json); unique weight bytes per launch = 22.1e6",
                "avg_launch_us": round(avg_s * 1e6, 2), "launches": launches, "launches_per_step": round(launches / K, 2),
                "flop_per_launch": flop_per_launch,
                "path_achieved": round(path, 2), "path_frac": round(path / PEAK_FP32_MFMA_TFLOPS, 4),
                "note": "frac: rc_gemm_kernel alone (its algorithmic FLOPs / its HIP-event time); path_frac: whole frame incl. "
                        "the weight-streaming 16-row launches and the per-frame logic kernels"}

    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(sd, body, m)
        value = world * B * K / dt
        print(json.dumps({
            "metric": "body-frames/sec (sig_mp fwd + FK) at batch 256", "value": round(value, 1), "unit": "body-frames/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"synthetic 60 fps, 6 IMU + 33 keypoints, batch {B} x {K} frames per GPU, "
                                   f"confidence schedule '{args.conf}', seeded random weights (63.4 M params)",
                       "batch_per_gpu": B, "frames": K, "conf": args.conf, "parallelism": f"dp{world} (sequence sharding)"},
            "roofline": roof, "cpu_baseline": cpu}))
    if world > 1:
        torch.distributed.barrier()          # rank 0 may still be in its instrumented pass
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
