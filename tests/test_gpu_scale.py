"""Full-size checks of the HIP path through size-independent properties (BASELINE.json configs 2 and 4).

The oracle is too slow for 131k body-frames, so at full size the path is checked through: orthonormal outputs,
root == pelvis IMU, bitwise determinism, sharded == unsharded (the multi-GPU decomposition, run as two contexts on
one GPU), batch row == single-sequence run, plus an oracle spot check on sampled rows."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
t = torch.from_numpy


def _net(assets, B):
    from robustcap_amd.net.sig_mp import Net
    n = Net(body=assets["body"], batch=B)
    n.load_state_dict(assets["state_dict"])
    assert n.gemm_mode == (1 if B >= 48 else 0) == int(Net.default_gemm_mode(B))   # split-bf16 products from batch 48 up
    return n


def _inputs(assets, B, T, conf, seed):
    import bench
    return bench.make_inputs(assets["body"], B, T, conf, seed)


def _run(net, m, rows=slice(None), T=None, first_tran=True):
    sl = slice(0, T)
    net.gravityc = t(m["gravityc"][rows])
    return net.forward_sequence(t(m["j2dc"][rows, sl]), t(m["accc"][rows, sl]), t(m["oric"][rows, sl]),
                                first_tran=t(m["first_tran"][rows]) if first_tran else None, first_frame=not first_tran)


@pytest.fixture(scope="module")
def cfg2(synth_assets):
    B, T = 256, 512
    m = _inputs(synth_assets, B, T, "mixed", 2)
    net = _net(synth_assets, B)
    pose, tran = _run(net, m)
    torch.cuda.synchronize()
    return m, pose, tran


def test_config2_outputs_are_valid_rotations(cfg2):
    m, pose, tran = cfg2
    assert torch.isfinite(pose).all() and torch.isfinite(tran).all()
    R = pose.reshape(-1, 3, 3).double()
    eye = torch.eye(3, dtype=torch.float64, device=R.device)
    assert float((R @ R.transpose(1, 2) - eye).abs().max()) < 2e-5
    assert float((torch.linalg.det(R) - 1).abs().max()) < 2e-5
    assert torch.equal(pose[:, :, 0].cpu(), t(m["oric"][:, :, 5]))                     # pose[0] = Rcr (sig_mp.py:175)
    assert torch.equal(tran[:, 0].cpu(), t(m["first_tran"]))                           # frame 0 = first_tran (L222-223)
    regimes = m["conf"]
    assert (regimes >= 0.8).mean() > 0.2 and (regimes <= 0.7).mean() > 0.1             # the workload really is mixed


def test_config2_deterministic_and_shard_equivalent(cfg2, synth_assets):
    m, pose, tran = cfg2
    again = _net(synth_assets, 256)
    p2, t2 = _run(again, m)
    assert torch.equal(p2, pose) and torch.equal(t2, tran)                              # bitwise repeatable
    for a, b in ((0, 128), (128, 256)):                                                 # two "ranks" of 128 rows each
        shard = _net(synth_assets, b - a)
        shard.set_gemm_mode(True)                         # the product arithmetic of the 256-row context (128-row contexts default to fp32 MFMA)
        ps, ts = _run(shard, m, rows=slice(a, b))
        assert torch.equal(ps, pose[a:b]) and torch.equal(ts, tran[a:b])


def test_config2_rows_equal_single_sequence_runs_and_oracle(cfg2, synth_assets):
    from oracle import sig_mp_oracle as O
    m, pose, tran = cfg2
    Tc = 96
    for b in (0, 77, 255):
        one = _net(synth_assets, 1)
        one.set_gemm_mode(True)                           # the arithmetic of the 256-row context (batch-1 contexts default to fp32 MFMA)
        p1, t1 = _run(one, m, rows=slice(b, b + 1), T=Tc)
        assert torch.equal(p1[0], pose[b, :Tc]) and torch.equal(t1[0], tran[b, :Tc])    # row b == that sequence alone
    rows = [3, 200]
    ora = O.OracleNet(synth_assets["body"], batch=len(rows))
    ora.load_numpy_state_dict(synth_assets["state_dict"])
    ora.gravityc = t(m["gravityc"][rows])
    ob = O.OracleBody(synth_assets["body"])
    for i in range(64):
        p, tr = ora.forward_batch(t(m["j2dc"][rows, i]), t(m["accc"][rows, i]), t(m["oric"][rows, i]),
                                  t(m["first_tran"][rows]) if i == 0 else None)
        gp, gt = pose[rows, i].cpu(), tran[rows, i].cpu()
        assert float((gt - tr).abs().max()) <= 1e-4, i
        assert float(O.rotation_angle_deg(gp, p).max()) <= 0.1, i
        assert float((ob.forward_kinematics(gp, gt)[1] - ob.forward_kinematics(p, tr)[1]).abs().max()) <= 1e-4, i


def _trace8(ora, conf_lo=0.7, conf_hi=0.8):
    """The oracle's per-row branch record of its last frame in rc_get_trace's layout: {regime, rnn4 steps, rnn6 steps, floor samples
    held, reach fired, used velocity branch, stance foot, jump reset}."""
    tr = ora.trace
    c = tr["c"].double()
    regime = (c > conf_lo).long() + (c >= conf_hi).long()
    cols = [regime, tr["n4"].long(), tr["n6"].long(), tr["n_floor"].long(), tr["reach"].long(), tr["use_vel"].long(), tr["foot"].long(), tr["far"].long()]
    return torch.stack([x.reshape(-1) for x in cols], 1)


def test_config2_two_rows_against_the_oracle_over_all_512_frames(cfg2, synth_assets):
    """Round-5 review: the oracle comparison of the full-size run covered 64 of its 512 frames. Here every frame of two rows of the
    256 x 512 run (split products, wavefront engine, rows lagging through their occlusions) against the CPU restatement stepping the
    same two sequences frame by frame: the only place where long-horizon drift AT BATCH is compared with the oracle, not with itself."""
    from oracle import sig_mp_oracle as O
    m, pose, tran = cfg2
    rows = [9, 40]
    ora = O.OracleNet(synth_assets["body"], batch=len(rows))
    ora.load_numpy_state_dict(synth_assets["state_dict"])
    ora.gravityc = t(m["gravityc"][rows])
    ob = O.OracleBody(synth_assets["body"])
    T = pose.shape[1]
    worst = [0.0, 0.0, 0.0]
    for i in range(T):
        p, tr = ora.forward_batch(t(m["j2dc"][rows, i]), t(m["accc"][rows, i]), t(m["oric"][rows, i]),
                                  t(m["first_tran"][rows]) if i == 0 else None)
        gp, gt = pose[rows, i].cpu(), tran[rows, i].cpu()
        worst[0] = max(worst[0], float((gt - tr).abs().max()))
        worst[1] = max(worst[1], float(O.rotation_angle_deg(gp, p).max()))
        if i % 8 == 0 or i == T - 1:
            worst[2] = max(worst[2], float((ob.forward_kinematics(gp, gt)[1] - ob.forward_kinematics(p, tr)[1]).abs().max()))
    assert worst[0] <= 1e-4 and worst[1] <= 0.1 and worst[2] <= 1e-4, worst
    conf = m["conf"][rows]
    assert all((c <= 0.7).mean() > 0.15 and (c >= 0.8).mean() > 0.3 for c in conf)       # both rows really change regime on the way


def test_config4_rows_against_the_oracle_at_batch_1024(synth_assets):
    """BASELINE config 4 at its own size (batch 1024, occlusion-masked keypoints): four sampled rows x 48 frames of the full run against
    the CPU restatement (1e-4 m / 0.1 deg, joints 1e-4 m), and the branch record of their last frame."""
    from oracle import sig_mp_oracle as O
    B, T = 1024, 48
    m = _inputs(synth_assets, B, T, "occ", 4)
    net = _net(synth_assets, B)
    pose, tran = _run(net, m, first_tran=False)
    torch.cuda.synchronize()
    low_any = (m["conf"] <= 0.7).any(1) & (m["conf"] > 0.7).any(1)                       # rows that enter or leave an occlusion inside the call
    rows = [int(r) for r in np.flatnonzero(low_any)[[0, 5, -7, -1]]]
    ora = O.OracleNet(synth_assets["body"], batch=len(rows))
    ora.load_numpy_state_dict(synth_assets["state_dict"])
    ora.gravityc = t(m["gravityc"][rows])
    ob = O.OracleBody(synth_assets["body"])
    for i in range(T):
        p, tr = ora.forward_batch(t(m["j2dc"][rows, i]), t(m["accc"][rows, i]), t(m["oric"][rows, i]), None, i == 0)
        gp, gt = pose[rows, i].cpu(), tran[rows, i].cpu()
        assert float((gt - tr).abs().max()) <= 1e-4, i
        assert float(O.rotation_angle_deg(gp, p).max()) <= 0.1, i
        assert float((ob.forward_kinematics(gp, gt)[1] - ob.forward_kinematics(p, tr)[1]).abs().max()) <= 1e-4, i
    got = net.get_trace()[rows].long()
    want = _trace8(ora)
    assert torch.equal(got, want), (got.tolist(), want.tolist())


def test_config4_occluded_batch_1024(synth_assets):
    B, T = 1024, 48
    m = _inputs(synth_assets, B, T, "occ", 4)
    net = _net(synth_assets, B)
    pose, tran = _run(net, m, first_tran=False)
    assert torch.isfinite(pose).all() and torch.isfinite(tran).all()
    tr = net.get_trace()
    low = t((m["conf"][:, T - 1] <= 0.7))
    assert torch.equal(tr[:, 0] == 0, low)                                               # regime flags follow the input
    assert int(tr[low][:, 1].min()) == 1 and int(tr[low][:, 2].min()) == 1               # occluded rows ran the updater
    half = _net(synth_assets, 512)
    ph, th = _run(half, m, rows=slice(512, 1024), first_tran=False)
    assert torch.equal(ph, pose[512:]) and torch.equal(th, tran[512:])


def test_strong_split_over_two_ranks_equals_one_rank():
    """bench.py --scaling strong / dist.shard_range: 2 ranks (sharing this box's one GPU, gloo collective) run their row
    blocks of ONE 37-body batch (19 + 18 rows), rank 0 gathers -- bitwise the 1-rank result; and the bench itself runs
    in that mode under the launcher the driver uses."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def port():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            return str(s.getsockname()[1])
    env = dict(os.environ, RC_DIST_SHARE_DEVICE="1", RC_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1"]
    r = subprocess.run(base + ["--master-port", port(), os.path.join(root, "tools", "strong_split_check.py"), "37", "40"],
                       capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["bitwise_equal"] and d["world"] == 2 and d["blocks"] == [[0, 19], [19, 37]]
    r = subprocess.run(base + ["--master-port", port(), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "4",
                               "--scaling", "strong", "--batch", "37", "--no-variants"],
                       capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["bodies_total"] == 37 and d["config"]["batch_per_gpu"] == 19
    assert d["value"] > 0 and d["cpu_baseline"] is None
    # weak scaling with the variants, as the driver launches it for N > 1 (here 2 ranks x 48 bodies on the one GPU)
    r = subprocess.run(base + ["--master-port", port(), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "4", "--batch", "48"],
                       capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["bodies_total"] == 96 and d["variants"]["high"]["value"] > 0
    assert d["roofline"] is not None and d["cpu_baseline"] is None
    # `python bench.py --gpus 2` with NO launcher: the bench starts its own ranks (torch.distributed.run on a free port), the line
    # says n_gpus 2, every rank answered the collective, and the strong split of the headline batch sits beside the weak figure
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "4", "--batch", "48",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=root,
                       env={k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["rccl"]["ranks_seen"] == 2 and d["config"]["bodies_total"] == 96
    assert d["variants"]["strong"]["value"] > 0 and "48 bodies in total over 2 ranks" in d["variants"]["strong"]["workload"]
