"""Harness (evaluate.py counterpart): camera input prep kernel, row batching of (sequence, camera), metrics."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
t = torch.from_numpy


def test_camera_inputs_match_reference_formulas(synth_assets):
    from robustcap_amd import synth
    from robustcap_amd.evaluate import camera_inputs
    ds = synth.make_dataset(5, 1, 16, synth_assets["body"], n_cam=2)
    K, Tcw = t(ds["cam_K"][0][1]), t(ds["cam_T"][0][1])
    j2dc, accc, oric, g = camera_inputs(ds["joint2d_mp"][0][1], ds["imu_acc"][0], ds["imu_ori"][0], K, Tcw)
    kp = t(ds["joint2d_mp"][0][1]).clone()
    kp[..., 0] *= 1920
    kp[..., 1] *= 1080
    ones = torch.cat((kp[..., :2], torch.ones_like(kp[..., :1])), -1)
    ref = (K.inverse() @ ones.unsqueeze(-1)).squeeze(-1)                           # evaluate.py:70-71
    ref[..., -1] = kp[..., -1]                                                      # evaluate.py:72
    assert float((j2dc.cpu() - ref).abs().max()) <= 2e-6
    assert float((oric.cpu() - Tcw[:3, :3] @ t(ds["imu_ori"][0])).abs().max()) <= 1e-6        # evaluate.py:38
    assert float((accc.cpu() - t(ds["imu_acc"][0]) @ Tcw[:3, :3].T).abs().max()) <= 1e-5      # evaluate.py:39
    assert float((g - Tcw[:3, :3] @ torch.tensor([0.0, -1.0, 0.0])).abs().max()) <= 1e-7      # evaluate.py:73


def test_batched_camera_inputs_equal_the_per_row_kernel(synth_assets):
    """rc_camera_inputs_rows (all rows, one launch, padded to Tmax) == rc_camera_inputs row by row, bit for bit."""
    from robustcap_amd import synth
    from robustcap_amd import evaluate as ev
    ds = synth.make_dataset(7, 3, 24, synth_assets["body"], n_cam=3)
    for k in ("pose", "tran", "imu_ori", "imu_acc"):
        ds[k][2] = ds[k][2][:17]                                                     # ragged
    ds["joint2d_mp"][2] = ds["joint2d_mp"][2][:, :17]
    rows = [(0, 0), (2, 1), (1, 2), (2, 2), (0, 1)]                                  # any order, any subset
    j2d, acc, ori, grav = ev.camera_inputs_rows(ds, rows, 24)
    for r, (i, j) in enumerate(rows):
        k, a, o, g = ev.camera_inputs(ds["joint2d_mp"][i][j], ds["imu_acc"][i], ds["imu_ori"][i], ds["cam_K"][i][j], ds["cam_T"][i][j])
        T = k.shape[0]
        assert torch.equal(j2d[r, :T], k) and torch.equal(acc[r, :T], a) and torch.equal(ori[r, :T], o) and torch.equal(grav[r], g)
        assert float(j2d[r, T:].abs().sum()) == 0.0 and float(acc[r, T:].abs().sum()) == 0.0
        if T < 24:
            assert torch.equal(ori[r, T:], torch.eye(3, device=ori.device).expand(24 - T, 6, 3, 3))


def test_run_dataset_rows_equal_direct_runs_and_metrics(synth_assets):
    from robustcap_amd import synth
    from robustcap_amd import evaluate as ev
    from robustcap_amd.body import ParametricModel
    from robustcap_amd.net.sig_mp import Net
    body, sd = synth_assets["body"], synth_assets["state_dict"]
    ds = synth.make_dataset(6, 2, 40, body, n_cam=2)
    ds["pose"][1], ds["tran"][1] = ds["pose"][1][:30], ds["tran"][1][:30]          # ragged: sequence 1 is shorter
    ds["imu_ori"][1], ds["imu_acc"][1] = ds["imu_ori"][1][:30], ds["imu_acc"][1][:30]
    ds["joint2d_mp"][1] = ds["joint2d_mp"][1][:, :30]
    res = ev.run_dataset(ds, sd, body)
    assert sorted(res) == [(0, 0), (0, 1), (1, 0), (1, 1)] and res[(1, 1)][0].shape == (30, 24, 3, 3)
    for (i, j) in ((0, 1), (1, 0)):                                                # one row run alone, like evaluate.py's loop
        k, a, o, g = ev.camera_inputs(ds["joint2d_mp"][i][j], ds["imu_acc"][i], ds["imu_ori"][i], ds["cam_K"][i][j], ds["cam_T"][i][j])
        net = Net(body=body, batch=1)
        net.load_state_dict(sd)
        net.gravityc = g
        ft = ev.labels(ds, i, j)[1][0]
        p, tr = net.forward_sequence(k[None], a[None], o[None], first_tran=ft[None])
        assert torch.equal(p[0].cpu(), res[(i, j)][0]) and torch.equal(tr[0].cpu(), res[(i, j)][1])
    model = ParametricModel(body=body)
    pt, tt = ev.labels(ds, 0, 0)
    z = ev.joint_errors(model, pt, tt, pt, tt)
    assert z["mpjpe_smpl24_m"] == 0.0 and z["root_error_m"] == 0.0 and z["global_angle_deg"] < 1e-3
    shifted = ev.joint_errors(model, pt, tt + torch.tensor([0.1, 0.0, 0.0]), pt, tt)
    assert abs(shifted["root_error_m"] - 0.1) < 1e-5 and shifted["mpjpe_smpl24_m"] < 1e-5
    Rz = torch.tensor([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    rot = pt.clone()
    rot[:, 0] = Rz @ rot[:, 0]                                                    # rigidly rotated prediction
    e = ev.joint_errors(model, rot, tt, pt, tt)
    assert e["mpjpe_smpl24_m"] > 0.05 and e["pa_mpjpe_smpl24_m"] < 1e-5           # Procrustes removes it
    per_row, mean = ev.evaluate(ds, sd, body)
    assert len(per_row) == 4 and np.isfinite(list(mean.values())).all()


def test_run_dataset_with_smplify_rows_in_parallel(synth_assets):
    """evaluate.py:86-90 over all rows: several rows' optimisers in flight (own context + stream per host thread) give the
    same result as one after another, and every row was optimised."""
    from robustcap_amd import synth
    from robustcap_amd import evaluate as ev
    body, sd, gmm = synth_assets["body"], synth_assets["state_dict"], synth.make_gmm(3)
    ds = synth.make_dataset(8, 2, 48, body, n_cam=3, conf="high")
    out = []
    for workers in (1, 3):
        info = {}
        res = ev.run_dataset(ds, sd, body, run_smplify=True, gmm=gmm, smplify_info=info, smplify_workers=workers)
        assert len(info) == 6 and all(v["n_eval"] >= 1 for v in info.values())
        out.append((res, info))
    for key in out[0][0]:
        assert torch.equal(out[0][0][key][0], out[1][0][key][0]) and torch.equal(out[0][0][key][1], out[1][0][key][1]), key
        assert out[0][1][key]["n_eval"] == out[1][1][key]["n_eval"]
    plain = ev.run_dataset(ds, sd, body)
    changed = [float((plain[k][1] - out[0][0][k][1]).abs().max()) for k in plain if out[0][1][k]["status"] == 1]
    assert changed and max(changed) > 0.0                                          # the optimiser moved the optimised rows


def test_full_mesh_and_cal_mpjpe(synth_assets):
    """rc_body_mesh == the oracle's skinning on every vertex (the oracle is pinned to the reference on sampled
    vertices); cal_mpjpe semantics: zero for identical poses, PVE > 0 and PA-MPJPE ~ 0 for a rigid rotation."""
    import os
    from oracle import sig_mp_oracle as O
    from robustcap_amd import evaluate as ev
    from robustcap_amd.body import ParametricModel
    body = synth_assets["body"]
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ops.npz"))
    model = ParametricModel(body=body)
    pose, tran = t(g["fk_pose"]), t(g["fk_tran"])
    vert = model.forward_mesh(pose, tran)
    assert vert.shape == (pose.shape[0], 6890, 3)
    ids = [int(i) for i in g["fk_vert_extra_ids"]]
    assert float((vert[:, ids].cpu() - t(g["fk_vert_extra"])).abs().max()) <= 2e-6           # reference capture
    full = O.OracleBody(body, vertex_ids=range(6890)).forward_kinematics(pose, tran)[2]
    assert float((vert.cpu() - full).abs().max()) <= 2e-6
    z = ev.cal_mpjpe(model, pose, pose, cal_pampjpe=True)
    assert z[0] == 0.0 and z[1] == 0.0 and z[2] < 1e-6
    rot = pose.clone()
    rot[:, 0] = torch.tensor([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]]) @ rot[:, 0]
    e = ev.cal_mpjpe(model, rot, pose, cal_pampjpe=True)
    assert e[0] > 0.05 and e[1] > 0.05 and e[2] < 1e-5
    Jr = np.zeros((17, 6890), np.float32)
    Jr[np.arange(17), np.arange(17) * 400] = 1.0                                              # a stand-in regressor
    r = ev.cal_mpjpe(model, rot, pose, j_regressor=Jr)
    assert len(r) == 2 and r[0] > 0.0
    picked = vert[:, np.arange(14) * 400] - tran.to(vert.device)[:, None]                  # what that regressor selects
    rv = model.forward_mesh(rot, tran)[:, np.arange(14) * 400] - tran.to(vert.device)[:, None]
    want = float(((picked - picked[:, :1]) - (rv - rv[:, :1])).norm(dim=2).mean())
    assert abs(r[0] - want) <= 1e-6


@pytest.mark.parametrize("name", ["near", "far", "same"])
def test_cal_mpjpe_matches_reference_capture(name, synth_assets):
    """rc_mesh_metrics against the reference's own cal_mpjpe (tests/golden/metrics.npz): per-frame MPJPE, PVE and
    PA-MPJPE, and the three means."""
    import os
    from robustcap_amd import evaluate as ev
    from robustcap_amd import synth
    from robustcap_amd.body import ParametricModel
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "metrics.npz"))
    model = ParametricModel(body=synth_assets["body"])
    Jr = synth.make_j_regressor(4)
    pose = t(g["pose_gt"] if name == "same" else g["pose_" + name])
    out = ev.cal_mpjpe(model, pose, t(g["pose_gt"]), j_regressor=Jr, cal_pampjpe=True)
    assert np.allclose(out, g["cal_" + name], atol=3e-6)
    assert ev.cal_mpjpe(model, pose, t(g["pose_gt"]), j_regressor=Jr) == out[:2]
    per_frame, _ = model.mesh_metrics(pose, t(g["pose_gt"]))
    pf = per_frame.cpu().numpy()
    assert np.abs(pf[:, 0] - g["frame_mpjpe_" + name]).max() <= 3e-6
    assert np.abs(pf[:, 1] - g["frame_pve_" + name]).max() <= 3e-6
    assert np.abs(pf[:, 2] - g["frame_pa_" + name]).max() <= 3e-6


def test_procrustes_and_position_error_match_reference_capture(synth_assets):
    """The Procrustes stage on the reference's raw point sets (incl. mirrored sets: the det < 0 branch of
    utils.py:171-174), PositionErrorEvaluator, and the fused kernel against the oracle on a second pose pair."""
    import os
    from oracle import metrics_oracle as M
    from robustcap_amd import synth
    from robustcap_amd.body import ParametricModel, position_error, reconstruction_error
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "metrics.npz"))
    assert abs(position_error(t(g["pos_a"]), t(g["pos_b"])) - float(g["pos_err"])) <= 1e-6
    err = reconstruction_error(t(g["pa_S1"]), t(g["pa_S2"])).cpu().numpy()
    assert np.abs(err - g["pa_err"]).max() <= 1e-5 * np.abs(g["pa_err"]).max()
    assert reconstruction_error(torch.zeros(0, 14, 3), torch.zeros(0, 14, 3)).shape == (0,)
    body = synth_assets["body"]
    model = ParametricModel(body=body)
    model.set_regressor(synth.make_j_regressor(4), 14)
    gt = t(g["pose_gt"])
    Rz = torch.tensor([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    rot = gt.clone()
    rot[:, 0] = Rz @ rot[:, 0]
    pf, mean = model.mesh_metrics(rot, gt)
    assert mean[0] > 1e-3 and mean[2] < 1e-6                                                 # a rigid rotation is aligned away
    want = M.frame_metrics(body, synth.make_j_regressor(4), t(g["pose_far"]), rot)
    pf, _ = model.mesh_metrics(t(g["pose_far"]), rot)
    pf = pf.cpu().numpy()
    for c in range(3):
        assert np.abs(pf[:, c] - want[c]).max() <= 3e-6


def test_dataset_metrics_in_one_call_equal_the_row_by_row_loop(synth_assets):
    """evaluate.dataset_metrics (all rows' frames in ONE rc_mesh_metrics call, ground-truth rotations built once per
    sequence) == cal_mpjpe row by row as the reference's loop runs it (evaluate.py:95-100, 120-133)."""
    from robustcap_amd import synth
    from robustcap_amd import evaluate as ev
    from robustcap_amd.body import ParametricModel
    body = synth_assets["body"]
    ds = synth.make_dataset(8, 3, 20, body, n_cam=2)
    for k in ("pose", "tran", "imu_ori", "imu_acc"):
        ds[k][1] = ds[k][1][:13]                                                     # ragged
    ds["joint2d_mp"][1] = ds["joint2d_mp"][1][:, :13]
    g = torch.Generator().manual_seed(3)
    res = {}
    for (i, j) in ev.rows_of(ds):                                                    # "predictions": perturbed labels
        pt, tt = ev.labels(ds, i, j)
        noise = 0.1 * torch.randn(pt.shape[0], 24, 3, generator=g)
        from robustcap_amd import body as B
        res[(i, j)] = (pt @ B.axis_angle_to_rotation_matrix(noise.view(-1, 3)).view(-1, 24, 3, 3).cpu(), tt)
    model = ParametricModel(body=body)
    model.set_regressor(synth.make_j_regressor(4), 14)
    per_row, mean = ev.dataset_metrics(model, ds, res)
    assert sorted(per_row) == sorted(res)
    for (i, j), (pose, _) in res.items():
        pt, _ = ev.labels(ds, i, j)
        one = model.mesh_metrics(pose, pt)[1]
        assert max(abs(a - b) for a, b in zip(one, per_row[(i, j)])) < 1e-6
    assert all(v > 1e-3 for v in mean)


def test_config3_at_its_own_size(synth_assets):
    """BASELINE config 3 (the AIST++ evaluation: every (sequence, camera) row through the harness, evaluate.py:20-117) at the
    size tools/config3_eval.py quotes -- 8 sequences x 9 cameras x 600 frames = 72 rows -- with smplify on:
      * size-independent properties of the net's outputs (finite, orthonormal rotations, the IMU root in place),
      * a row of the batch == that row run alone (bitwise, same product arithmetic),
      * an oracle spot check (2 rows x 64 frames, 1e-4 m / 0.1 deg),
      * the row blocks of 8 ranks (dist.shard_range) computed one after another == the unsharded run, bitwise,
      * every row refined by the batched optimiser, the metric call over all rows."""
    from oracle import sig_mp_oracle as O
    from robustcap_amd import dist as rdist
    from robustcap_amd import evaluate as ev
    from robustcap_amd import synth
    from robustcap_amd.body import ParametricModel
    from robustcap_amd.net.sig_mp import Net
    body, sd, gmm = synth_assets["body"], synth_assets["state_dict"], synth.make_gmm(3)
    n_seq, n_cam, T = 8, 9, 600
    ds = synth.make_dataset(21, n_seq, T, body, n_cam=n_cam, conf="mixed")
    rows = ev.rows_of(ds)
    assert len(rows) == 72
    nets = {}
    plain = ev.run_dataset(ds, sd, body, nets=nets)
    assert sorted(plain) == sorted(rows)
    split = Net.default_gemm_mode(len(rows))
    assert split                                                                   # 72 rows: split-bf16 products
    # ---- properties at full size
    P = torch.stack([plain[k][0] for k in rows])                                   # [72, 600, 24, 3, 3]
    Tr = torch.stack([plain[k][1] for k in rows])
    assert P.shape == (72, T, 24, 3, 3) and torch.isfinite(P).all() and torch.isfinite(Tr).all()
    eye = torch.eye(3)
    assert float((P.transpose(-1, -2) @ P - eye).abs().max()) < 1e-4
    assert float((torch.linalg.det(P) - 1).abs().max()) < 1e-4
    for (i, j) in ((0, 0), (5, 7)):                                                # pose[0] = the pelvis IMU in the camera frame (sig_mp.py:175)
        o = ev.camera_inputs(ds["joint2d_mp"][i][j], ds["imu_acc"][i], ds["imu_ori"][i], ds["cam_K"][i][j], ds["cam_T"][i][j])[2]
        assert torch.equal(plain[(i, j)][0][:, 0], o[:, 5].cpu())
    # ---- a row of the batch == that row alone (same arithmetic), and the oracle on its first 64 frames
    for (i, j) in ((2, 3), (7, 8)):
        k, a, o, g = ev.camera_inputs(ds["joint2d_mp"][i][j], ds["imu_acc"][i], ds["imu_ori"][i], ds["cam_K"][i][j], ds["cam_T"][i][j])
        net = Net(body=body, batch=1)
        net.load_state_dict(sd)
        net.set_gemm_mode(split)
        net.gravityc = g
        ft = ev.labels(ds, i, j)[1][0]
        p, tr = net.forward_sequence(k[None], a[None], o[None], first_tran=ft[None])
        assert torch.equal(p[0].cpu(), plain[(i, j)][0]) and torch.equal(tr[0].cpu(), plain[(i, j)][1])
        ref = O.OracleNet(body, batch=1)
        ref.load_numpy_state_dict(sd)
        ref.gravityc = g
        kc, ac, oc = k.cpu(), a.cpu(), o.cpu()
        for f in range(64):
            rp, rt = ref.forward_online(kc[f], ac[f], oc[f], ft if f == 0 else None)
            assert float((rt - plain[(i, j)][1][f]).abs().max()) < 1e-4, (i, j, f)
            cosang = ((rp.transpose(-1, -2) @ plain[(i, j)][0][f]).diagonal(dim1=-2, dim2=-1).sum(-1) - 1) / 2
            assert float(torch.rad2deg(torch.acos(cosang.clamp(-1, 1))).max()) < 0.1, (i, j, f)
    # ---- the partition of 8 ranks, block by block on this one device == unsharded (what shard_rows + one gather give on 8 GPUs)
    for rank in range(8):
        a_, b_ = rdist.shard_range(len(rows), rank, 8)
        assert b_ - a_ == 9
        part = ev.run_dataset(ds, sd, body, rows=rows[a_:b_], nets=nets, gemm_mode=split)
        for key in rows[a_:b_]:
            assert torch.equal(part[key][0], plain[key][0]) and torch.equal(part[key][1], plain[key][1]), (rank, key)
    # ---- smplify over all rows (one lock-step batch) and the metrics of the refined result
    info = {}
    refined = ev.run_dataset(ds, sd, body, run_smplify=True, gmm=gmm, smplify_info=info, nets=nets)
    assert len(info) == 72 and all(v["n_eval"] >= 1 for v in info.values())
    done = [k for k in rows if info[k]["status"] == 1]
    assert len(done) >= 36                                                          # (rows the pre-check rejects stay as they are)
    assert all(torch.isfinite(refined[k][0]).all() and torch.isfinite(refined[k][1]).all() for k in rows)
    assert max(float((refined[k][1] - plain[k][1]).abs().max()) for k in done) > 0.0
    R = torch.stack([refined[k][0] for k in done])
    assert float((R.transpose(-1, -2) @ R - eye).abs().max()) < 1e-4
    model = ParametricModel(body=body)
    model.set_regressor(synth.make_j_regressor(4), 14)
    per_row, mean = ev.dataset_metrics(model, ds, refined)
    assert len(per_row) == 72 and np.isfinite(np.asarray(mean)).all()
