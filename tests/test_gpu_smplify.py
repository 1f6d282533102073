"""smplify optimiser on the GPU against the reference capture and the oracle.

tests/golden/smplify.npz holds, from the reference itself (oracle/capture_smplify.py): loss and gradient of the
closure at one point (ev_*) and one full smplify_runner call (run_*)."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import sig_mp_oracle as O
from oracle import smplify_oracle as S
from robustcap_amd import config as C
from robustcap_amd import synth

pytestmark = pytest.mark.gpu
t = torch.from_numpy


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "smplify.npz"))


@pytest.fixture(scope="module")
def runner(synth_assets):
    from robustcap_amd.smplify import TemporalSMPLify
    return TemporalSMPLify(body=synth_assets["body"], gmm=synth.make_gmm(3))


def _imu_aa(ori):
    return O.rotation_matrix_to_axis_angle(ori.reshape(-1, 3, 3)).reshape(ori.shape[0], 18)


def test_closure_matches_reference_capture(g, runner):
    """Loss to 1e-5 relative and gradient to 1e-4 of its scale against the reference's own autograd through the
    6890-vertex mesh (the same bars tests/test_smplify_oracle.py holds the oracle to)."""
    loss, gp, gt = runner.loss_and_grad(t(g["ev_pose"]), t(g["ev_tran"]), t(g["ev_kp"]), t(g["ev_ref3d"]), _imu_aa(t(g["ev_imu_ori"])),
                                        t(g["ev_K"]))
    assert abs(loss - float(g["ev_loss"])) <= 1e-5 * abs(float(g["ev_loss"]))
    gs = max(np.abs(g["ev_grad_pose"]).max(), np.abs(g["ev_grad_tran"]).max())
    assert float((gp.cpu() - t(g["ev_grad_pose"])).abs().max()) <= 1e-4 * gs
    assert float((gt.cpu() - t(g["ev_grad_tran"])).abs().max()) <= 1e-4 * gs


def test_use_head_variant_matches_reference_capture(g, synth_assets):
    """use_head=True (temporal_smplify.py:93-94, the TotalCapture evaluation): only landmarks {31, 32} ignored."""
    from robustcap_amd.smplify import TemporalSMPLify
    r = TemporalSMPLify(body=synth_assets["body"], gmm=synth.make_gmm(3), use_head=True)
    loss, gp, gt = r.loss_and_grad(t(g["ev_pose"]), t(g["ev_tran"]), t(g["ev_kp"]), t(g["ev_ref3d"]), _imu_aa(t(g["ev_imu_ori"])), t(g["ev_K"]))
    assert abs(loss - float(g["evh_loss"])) <= 1e-5 * abs(float(g["evh_loss"]))
    gs = max(np.abs(g["evh_grad_pose"]).max(), np.abs(g["evh_grad_tran"]).max())
    assert float((gp.cpu() - t(g["evh_grad_pose"])).abs().max()) <= 1e-4 * gs
    assert float((gt.cpu() - t(g["evh_grad_tran"])).abs().max()) <= 1e-4 * gs
    pose = S.batch_rodrigues(t(g["ev_pose"]).view(-1, 3)).view(-1, 24, 3, 3)
    res = r.get_fitting_loss(pose, t(g["ev_tran"]), t(g["ev_kp"]), t(g["ev_K"]))
    assert float((res.cpu() - t(g["evh_residual"])).abs().max()) <= 1e-4 * float(g["evh_residual"].max())
    r.set_use_head(False)                                          # back to the default mask on the same context
    loss0, _, _ = r.loss_and_grad(t(g["ev_pose"]), t(g["ev_tran"]), t(g["ev_kp"]), t(g["ev_ref3d"]), _imu_aa(t(g["ev_imu_ori"])), t(g["ev_K"]))
    assert abs(loss0 - float(g["ev_loss"])) <= 1e-5 * abs(float(g["ev_loss"]))


def test_shape_variant_matches_reference_capture(g, synth_assets):
    """shape=... (temporal_smplify.py:84-86,158-159,211-216): closure loss / gradient and the residual on the shaped body
    against the reference's own autograd; ref3d stays the mean-shape landmarks the reference preserves (L112)."""
    from robustcap_amd.smplify import TemporalSMPLify
    beta = t(g["evs_beta"])
    r = TemporalSMPLify(body=synth_assets["body"], gmm=synth.make_gmm(3), shape=beta)
    loss, gp, gt = r.loss_and_grad(t(g["ev_pose"]), t(g["ev_tran"]), t(g["ev_kp"]), t(g["ev_ref3d"]), _imu_aa(t(g["ev_imu_ori"])), t(g["ev_K"]))
    assert abs(loss - float(g["evs_loss"])) <= 1e-5 * abs(float(g["evs_loss"]))
    gs = max(np.abs(g["evs_grad_pose"]).max(), np.abs(g["evs_grad_tran"]).max())
    assert float((gp.cpu() - t(g["evs_grad_pose"])).abs().max()) <= 1e-4 * gs
    assert float((gt.cpu() - t(g["evs_grad_tran"])).abs().max()) <= 1e-4 * gs
    assert abs(float(g["evs_loss"]) - float(g["ev_loss"])) > 1e-3 * abs(float(g["ev_loss"]))        # the shape matters
    pose = S.batch_rodrigues(t(g["ev_pose"]).view(-1, 3)).view(-1, 24, 3, 3)
    res = r.get_fitting_loss(pose, t(g["ev_tran"]), t(g["ev_kp"]), t(g["ev_K"]))
    assert float((res.cpu() - t(g["evs_residual"])).abs().max()) <= 1e-4 * float(g["evs_residual"].max())
    # the runner with shape=: optimises on the shaped body, keeps the mean-shape preserved landmarks; back to None afterwards
    from robustcap_amd.smplify import smplify_runner
    T = int(g["ev_T"])
    p1, t1, u1 = smplify_runner(pose, t(g["ev_tran"]), t(g["ev_kp"]), t(g["ev_imu_ori"]), T, t(g["ev_K"]), lr=0.001, shape=beta.view(1, 10).expand(T, 10),
                                runner=r)
    p0, t0, u0 = smplify_runner(pose, t(g["ev_tran"]), t(g["ev_kp"]), t(g["ev_imu_ori"]), T, t(g["ev_K"]), lr=0.001, runner=r)
    assert u1 is not None and u0 is not None and float((t1 - t0).abs().max()) > 1e-6
    with pytest.raises(NotImplementedError):
        smplify_runner(pose, t(g["ev_tran"]), t(g["ev_kp"]), t(g["ev_imu_ori"]), T, t(g["ev_K"]), use_lbfgs=False, runner=r)


@pytest.mark.parametrize("T,seed", [(1, 5), (2, 6), (37, 7), (150, 8)])
def test_closure_matches_oracle(T, seed, synth_assets, runner):
    """Seeded poses away from the capture, including T=1 (no temporal terms), a ragged length, and one that spans three 64-frame
    blocks of the prior kernel (the last one partly filled; several mixtures chosen across the frames)."""
    body = synth_assets["body"]
    obody, prior = O.OracleBody(body), S.Prior(synth.make_gmm(3))
    rnd = lambda stream, *shape: synth.normal(seed, stream, int(np.prod(shape))).reshape(shape).astype(np.float32)
    bp = t(0.35 * rnd(0, T, 72))
    tr = t((np.array([0.1, -0.2, 3.0], np.float32) + 0.2 * rnd(1, T, 3)).astype(np.float32))
    K = torch.tensor([[600.0, 0.0, 320.0], [0.0, 600.0, 240.0], [0.0, 0.0, 1.0]])
    with torch.no_grad():
        pose = S.batch_rodrigues(bp.view(-1, 3)).view(T, 24, 3, 3)
        _, joint, vert = obody.forward_kinematics(pose, tr)
        mj = obody.landmarks(vert, joint)
        proj = (K @ (mj / mj[..., 2:]).unsqueeze(-1)).squeeze(-1)[..., :2]
    kp = torch.cat([proj + 25.0 * t(rnd(2, T, 33, 2)), t(synth.uniform01(seed, 3, T * 33).reshape(T, 33, 1))], dim=-1)
    ref3d = mj + 0.05 * t(rnd(4, T, 33, 3))
    imu_ori = S.batch_rodrigues(t(0.5 * rnd(5, T * 6, 3))).view(T, 6, 3, 3)
    conf = kp[:, :, 2].clone()
    conf[:, list(C.smplify_ignored_landmarks)] = 0.0
    a, b = bp.clone().requires_grad_(True), tr.clone().requires_grad_(True)
    want = S.fitting_loss(obody, prior, a, b, kp[:, :, :2], conf, K, ref3d, imu_ori)
    want.backward()
    loss, gp, gt = runner.loss_and_grad(bp, tr, kp, ref3d, _imu_aa(imu_ori), K)
    assert abs(loss - float(want.detach())) <= 2e-5 * abs(float(want.detach()))
    gs = float(max(a.grad.abs().max(), b.grad.abs().max()))
    assert float((gp.cpu() - a.grad).abs().max()) <= 2e-4 * gs
    assert float((gt.cpu() - b.grad).abs().max()) <= 2e-4 * gs


def test_runner_against_reference_run(g, runner, synth_assets):
    """Same bars as the oracle's own pin (tests/test_smplify_oracle.py): L-BFGS trajectories separate after a few
    evaluations in float32, so the end state is compared through the pre-check, the evaluation budget, the update mask
    and the size of the improvement."""
    from robustcap_amd.smplify import smplify_runner
    T = int(g["run_T"])
    pose, tran, update = smplify_runner(t(g["run_pose0"]), t(g["run_tran0"]), t(g["run_kp"]), t(g["run_imu_ori"]), T, t(g["run_K"]),
                                        lr=0.001, runner=runner)
    info = runner.last_info
    ref = g["run_closure_losses"]
    assert info["status"] == 1 and info["n_eval"] == len(ref) == 26
    assert abs(info["first_loss"] - ref[0]) <= 1e-5 * ref[0]
    assert info["final_loss"] < 0.45 * info["first_loss"] and ref[-1] < 0.45 * ref[0]
    assert update is not None and np.array_equal(update.numpy(), g["run_update"])
    ob = O.OracleBody(synth_assets["body"])
    after = float(O.reprojection_residual(ob, pose, tran, t(g["run_kp"]), t(g["run_K"])).mean())
    before = float(g["run_loss_before"].mean())
    assert after < 0.4 * before and float(g["run_loss_after"].mean()) < 0.4 * before
    assert abs(after - float(g["run_loss_after"].mean())) < 0.15 * before            # same range as the reference's end state
    # rotation matrices out, orthonormal
    eye = torch.eye(3).expand(T, 24, 3, 3)
    assert float((pose @ pose.transpose(-1, -2) - eye).abs().max()) < 1e-5


def test_device_resident_lbfgs_against_the_host_formulation(g, runner, monkeypatch):
    """rc_smplify_run keeps the optimiser's vectors on the device and runs the two-loop recursion on coefficients (float64 inner
    products); RC_SMPLIFY_HOST_LBFGS=1 selects round 2's formulation (rc_lbfgs.h on host vectors, pinned against torch in
    tests/test_lbfgs_host.py). Same algorithm, different rounding: same evaluation budget, the same first closure value, end
    states in the same range."""
    from robustcap_amd.smplify import smplify_runner
    T = int(g["run_T"])
    args = (t(g["run_pose0"]), t(g["run_tran0"]), t(g["run_kp"]), t(g["run_imu_ori"]), T, t(g["run_K"]))
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RC_SMPLIFY_HOST_LBFGS", mode)
        smplify_runner(*args, lr=0.001, runner=runner)
        out[mode] = dict(runner.last_info)
    dev, host = out["0"], out["1"]
    assert dev["status"] == host["status"] == 1 and dev["n_eval"] == host["n_eval"] == 26 and dev["n_iter"] == host["n_iter"]
    assert dev["first_loss"] == host["first_loss"]                                  # the same closure on the same point
    assert abs(dev["final_loss"] - host["final_loss"]) < 0.1 * host["first_loss"]
    assert dev["final_loss"] < 0.45 * dev["first_loss"]


def test_lbfgs_history_slides_like_torch(g, runner, monkeypatch):
    """More iterations than curvature pairs fit (round-3 advice: the device-resident optimiser used to freeze its history once
    it was full). torch.optim.LBFGS drops the OLDEST pair (old_dirs.pop(0)); so does the host formulation (rc_lbfgs.h, pinned
    against torch in tests/test_lbfgs_host.py). With a history of 4 and 40 iterations both formulations spend the same
    evaluation budget and end in the same range, the run differs from the one with the full history (eviction is live), and
    more than 100 iterations (torch's history_size) run through."""
    from robustcap_amd.smplify import smplify_runner
    T = int(g["run_T"])
    args = (t(g["run_pose0"]), t(g["run_tran0"]), t(g["run_kp"]), t(g["run_imu_ori"]), T, t(g["run_K"]))
    out = {}
    for hist in ("4", "100"):
        for mode in ("0", "1"):
            monkeypatch.setenv("RC_LBFGS_HISTORY", hist)
            monkeypatch.setenv("RC_SMPLIFY_HOST_LBFGS", mode)
            runner.run(*args[:4], args[5], lr=0.001, max_iter=40)
            out[hist, mode] = dict(runner.last_info)
    for hist in ("4", "100"):
        dev, host = out[hist, "0"], out[hist, "1"]
        assert dev["status"] == host["status"] == 1 and dev["first_loss"] == host["first_loss"]
        assert abs(dev["n_eval"] - host["n_eval"]) <= 2 and abs(dev["n_iter"] - host["n_iter"]) <= 2, (hist, dev, host)
        assert dev["n_iter"] > 8                                         # well past a 4-pair history
        assert abs(dev["final_loss"] - host["final_loss"]) < 0.1 * host["first_loss"]
        assert dev["final_loss"] < 0.45 * dev["first_loss"]
    assert out["4", "0"]["final_loss"] != out["100", "0"]["final_loss"]  # the short history really evicts
    monkeypatch.delenv("RC_LBFGS_HISTORY")
    monkeypatch.setenv("RC_SMPLIFY_HOST_LBFGS", "0")
    runner.run(*args[:4], args[5], lr=0.001, max_iter=130)
    long_run = dict(runner.last_info)
    assert long_run["status"] == 1 and long_run["final_loss"] <= out["100", "0"]["final_loss"] * 1.05


def test_batched_rows_equal_one_row_at_a_time(runner, synth_assets):
    """rc_smplify_run_batch: the rows of an evaluation (evaluate.py:86-90) optimised in lock-step rounds on the device -- per row
    the same iterations, evaluations, losses and outputs as rc_smplify_run on that row alone; ragged lengths, different cameras,
    and a row the pre-check rejects (run.py:27-29) is copied through."""
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import smplify_bench as sb
    body = synth_assets["body"]
    rows = []
    for seed, T in ((11, 64), (23, 100), (31, 37), (47, 64), (53, 80)):
        pose0, tran0, kp, ori, K = sb.make_case(runner, body, T, seed=seed)
        K = K.clone()
        K[0, 0] += seed                                                   # a camera of its own per row
        rows.append((pose0, tran0, kp, ori, K))
    bad = list(rows[2])
    bad[2] = bad[2].clone()
    bad[2][..., :2] += 5.0e4                                              # keypoints far off: the robust residual saturates near 16,000 per landmark
    rows.append(tuple(bad))
    single = []
    for r in rows:
        p, tr, upd = runner.run(*r, lr=0.001, loss_threshold=1000.0)
        single.append((p.clone(), tr.clone(), None if upd is None else upd.clone(), dict(runner.last_info)))
    out = runner.run_batch(rows, lr=0.001, loss_threshold=1000.0)
    info = runner.last_batch_info
    assert single[-1][2] is None and out[-1][2] is None and info[-1]["status"] == 0
    assert torch.equal(out[-1][0], rows[-1][0].to(out[-1][0].device)) and torch.equal(out[-1][1], rows[-1][1].to(out[-1][1].device))
    for r in range(len(rows) - 1):
        a, b = single[r][3], info[r]
        assert b["status"] == 1 and (a["n_iter"], a["n_eval"]) == (b["n_iter"], b["n_eval"]), (r, a, b)
        assert a["first_loss"] == b["first_loss"] and abs(a["final_loss"] - b["final_loss"]) <= 1e-6 * abs(a["final_loss"]), (r, a, b)
        assert float((single[r][0] - out[r][0]).abs().max()) <= 1e-6 and float((single[r][1] - out[r][1]).abs().max()) <= 1e-6
        assert torch.equal(single[r][2], out[r][2])
    assert 20 <= info[0]["rounds"] <= 60                                   # lock-step: rounds of the longest row, not their sum


_SCHED_SCRIPT = r"""
import hashlib, json, os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tools"))
import torch
import smplify_bench as sb
from robustcap_amd import synth
from robustcap_amd.smplify import TemporalSMPLify
body = synth.make_body(1)
runner = TemporalSMPLify(body=body, gmm=synth.make_gmm(3))
rows = [sb.make_case(runner, body, T, seed=seed) for seed, T in ((11, 64), (23, 100), (31, 37))]
out = runner.run_batch(rows, lr=0.001)
h = hashlib.sha256()
for p, tr, upd in out:
    h.update(p.cpu().numpy().tobytes()); h.update(tr.cpu().numpy().tobytes()); h.update(upd.numpy().tobytes())
print(json.dumps({"sha": h.hexdigest(), "info": [[i["n_iter"], i["n_eval"], i["final_loss"]] for i in runner.last_batch_info]}))
"""


def test_rows_as_fibers_equal_rows_as_threads():
    """rc_smplify_run_batch drives every row's optimiser as a fiber of the caller's thread (makecontext / swapcontext); RC_SMPLIFY_THREADS=1
    selects a host thread per row instead. Same requests in the same rounds: outputs and optimiser records bitwise equal."""
    import json
    import subprocess
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    res = []
    for threads in ("0", "1"):
        env = dict(os.environ, RC_SMPLIFY_THREADS=threads)
        r = subprocess.run([sys.executable, "-c", _SCHED_SCRIPT, root], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert res[0] == res[1]
    assert all(i[1] >= 20 for i in res[0]["info"])


def test_runner_gate_and_errors(g, runner, synth_assets):
    from robustcap_amd import _lib
    from robustcap_amd.smplify import TemporalSMPLify, smplify_runner
    T = int(g["run_T"])
    args = (t(g["run_pose0"]), t(g["run_tran0"]), t(g["run_kp"]), t(g["run_imu_ori"]), T, t(g["run_K"]))
    gate = float(g["run_loss_before"][0].mean())
    pose, tran, update = smplify_runner(*args, lr=0.001, runner=runner, loss_threshold=gate - 1.0)   # run.py:27-29
    assert update is None and runner.last_info["status"] == 0
    assert torch.equal(pose, t(g["run_pose0"])) and torch.equal(tran, t(g["run_tran0"]))
    no_prior = TemporalSMPLify(body=synth_assets["body"])
    with pytest.raises(_lib.RobustcapLibraryError):
        smplify_runner(*args, runner=no_prior)
    with pytest.raises(_lib.RobustcapLibraryError):
        no_prior.loss_and_grad(t(g["ev_pose"]), t(g["ev_tran"]), t(g["ev_kp"]), t(g["ev_ref3d"]), torch.zeros(10, 18), t(g["ev_K"]))
    res = runner.get_fitting_loss(t(g["run_pose0"]), t(g["run_tran0"]), t(g["run_kp"]), t(g["run_K"]))
    assert float((res.cpu() - t(g["run_loss_before"])).abs().max()) <= 1e-4 * float(g["run_loss_before"].max())


def test_long_sequence_improves(runner, synth_assets):
    """A 600-frame sequence (the reference optimises whole sequences at once): the loss falls and the work buffers
    grow past the earlier calls."""
    body = synth_assets["body"]
    m = synth.make_motion(11, 1, 600, body, conf="high")
    K = torch.tensor([[600.0, 0.0, 320.0], [0.0, 600.0, 240.0], [0.0, 0.0, 1.0]])
    obody = O.OracleBody(body)
    pose = t(np.asarray(m["pose"][0], np.float32))
    tran = t(np.asarray(m["tran"][0], np.float32))
    with torch.no_grad():
        _, joint, vert = obody.forward_kinematics(pose, tran)
        mj = obody.landmarks(vert, joint)
        proj = (K @ (mj / mj[..., 2:]).unsqueeze(-1)).squeeze(-1)[..., :2]
    kp = torch.cat([proj, torch.full((600, 33, 1), 0.9)], dim=-1)
    noisy = O.axis_angle_to_rotation_matrix((0.05 * t(synth.normal(12, 0, 600 * 24 * 3).reshape(-1, 3)))).view(600, 24, 3, 3)
    pose0 = pose @ noisy
    gp, _, _ = obody.forward_kinematics(pose0, tran)
    imu_ori = gp[:, list(C.ji_mask)]
    p, tr, update = runner.run(pose0, tran + 0.02, kp, imu_ori, K, lr=0.001)
    info = runner.last_info
    assert info["status"] == 1 and info["n_eval"] == 26 and info["final_loss"] < 0.95 * info["first_loss"]
    assert update is not None and int(update.sum()) > 540                            # nearly every frame re-projects better
    before = float(O.reprojection_residual(obody, pose0, tran + 0.02, kp, K).mean())
    after = float(O.reprojection_residual(obody, p.cpu(), tr.cpu(), kp, K).mean())
    # the line search amplifies rounding noise after a few evaluations: the CPU oracle lands at 0.54 x,
    # this path between 0.55 x and 0.70 x depending on the last bits of the initial axis-angles -- the bar is 'clearly better'
    assert after < 0.75 * before
