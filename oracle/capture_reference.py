#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in this container (never on the GPU box).

TEST INFRASTRUCTURE. Imports /root/reference (read-only) with empty stand-in modules for packages that its
package __init__ files import eagerly but the hot path never executes (trimesh, pyrender, wandb, thop,
tensorboard, smplx; cv2 gets a Rodrigues that returns zeros -- its result only feeds the smplify prior /
imu-orientation terms, never the 'reprojection' output captured here). A synthetic SMPL-format pickle is
written to a temp cwd because net/sig_mp.py:19-20 loads ``models/SMPL_male.pkl`` at import.

What is stored: numbers only -- seeded inputs, the reference's outputs, a per-frame branch trace and the
per-frame outputs of each sub-net's ``linear2`` (forward hooks). Weights / body are NOT stored: fixtures hold
the generator seed + checksums (robustcap_amd.synth regenerates them bit-identically).

Usage:  python oracle/capture_reference.py            (writes tests/golden/: ops.npz, seq_*.npz, meta.json with sha256)
        python oracle/capture_reference.py NAME...    (only these scenarios)
        python oracle/capture_reference.py --check    (re-runs the reference on the STORED inputs: outputs must be bit-equal)
"""
import os
import pickle
import sys
import tempfile
import types

import numpy as np
import scipy.sparse
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from robustcap_amd import config as C  # noqa: E402
from robustcap_amd import synth  # noqa: E402
from oracle import _npz  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
WEIGHT_SEED, BODY_SEED = 0, 1


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    mod("trimesh"), mod("pyrender"), mod("wandb")
    mod("thop", clever_format=lambda *a, **k: None)
    mod("smplx", SMPL=object)
    mod("cv2", Rodrigues=lambda m: (np.zeros((3, 1), np.float32), None))
    tb = mod("tensorboard")
    import torch.utils
    tbu = mod("torch.utils.tensorboard", SummaryWriter=object)
    torch.utils.tensorboard = tbu
    del tb


def _write_body_pickle(path, body):
    V = body["v_template"].shape[0]
    kin = np.stack([np.where(body["parent"] < 0, 2 ** 32 - 1, body["parent"]).astype(np.int64),
                    np.arange(24, dtype=np.int64)])
    data = {
        "J_regressor": scipy.sparse.csc_matrix(np.asarray(body.get("J_regressor", np.zeros((24, V))), np.float64)),
        "weights": body["weights"].astype(np.float64),
        "posedirs": np.zeros((V, 3, 207), np.float32),
        "shapedirs": np.asarray(body.get("shapedirs", np.zeros((V, 3, 10))), np.float32),
        "v_template": body["v_template"].astype(np.float64),
        "J": body["J"].astype(np.float64),
        "f": np.zeros((1, 3), np.int64),
        "kintree_table": kin,
    }
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as f:
        pickle.dump(data, f)


def import_reference():
    body = synth.make_body(BODY_SEED)
    tmp = tempfile.mkdtemp(prefix="rc_ref_")
    _write_body_pickle(os.path.join(tmp, "models", "SMPL_male.pkl"), body)
    os.chdir(tmp)
    _install_stubs()
    sys.path.insert(0, REF)
    import articulate as art  # noqa
    import net.sig_mp as sig_mp  # noqa
    return art, sig_mp, body


# ------------------------------------------------------------------------------------------------------------
def rand_rot(seed, stream, n):
    aa = 2.5 * (synth.uniform01(seed, stream, n * 3).reshape(n, 3).astype(np.float64) - 0.5)
    return synth._rodrigues(aa).astype(np.float32)


def capture_ops(art, sig_mp, body):
    t = torch.from_numpy
    bm = sig_mp.body_model
    parent = bm.parent
    g = {}
    N = 32
    # r6d -> R (angular.py:249-264); last row degenerate (zero first vector -> NaN -> 0)
    r6d = (2 * synth.uniform01(11, 0, N * 24 * 6) - 1).reshape(N * 24, 6).astype(np.float32)
    r6d[-1, :3] = 0
    g["r6d_in"] = r6d
    g["r6d_out"] = art.math.r6d_to_rotation_matrix(t(r6d.copy())).numpy()
    # IK_R / FK_R (spatial.py:170-221)
    Rg = rand_rot(12, 0, N * 24).reshape(N, 24, 3, 3)
    g["ik_in"] = Rg
    g["ik_out"] = bm.inverse_kinematics_R(t(Rg.copy())).numpy()
    g["fkr_out"] = bm.forward_kinematics_R(t(g["ik_out"].copy())).numpy()
    # bone vectors of the rest pose (sig_mp.py:83-84) and bone-vector FK (spatial.py:126-145)
    j0 = bm.get_zero_pose_joint_and_vertex()[0]
    g["bone_rest"] = bm.joint_position_to_bone_vector(j0.unsqueeze(0))[0].numpy()
    pb = (synth.uniform01(13, 0, N * 72) - 0.5).reshape(N, 24, 3).astype(np.float32)
    g["bonefk_in"] = pb
    g["bonefk_out"] = bm.bone_vector_to_joint_position(t(pb.copy())).numpy()
    # full FK + LBS (model.py:209-241) and sync_mp3d (sig_mp.py:287-299, utils.py:129-135)
    pose = rand_rot(14, 0, N * 24).reshape(N, 24, 3, 3)
    tran = (4 * synth.uniform01(14, 1, N * 3) - 2).reshape(N, 3).astype(np.float32) + np.float32([0, 0, 5])
    grot, joint, vert = bm.forward_kinematics(t(pose.copy()), tran=t(tran.copy()), calc_mesh=True)
    extra = [0, 1, 777, 3000, 6889]
    g["fk_pose"], g["fk_tran"] = pose, tran
    g["fk_grot"], g["fk_joint"] = grot.numpy(), joint.numpy()
    g["fk_vert_mp"] = vert[:, list(C.mp_mask)].numpy()
    g["fk_vert_extra_ids"] = np.asarray(extra)
    g["fk_vert_extra"] = vert[:, extra].numpy()
    g["fk_j33"] = torch.stack([sig_mp.sync_mp3d(vert[i], joint[i]) for i in range(N)]).numpy()
    import utils as ref_utils
    assert torch.equal(ref_utils.sync_mp3d_from_smpl(vert, joint), t(g["fk_j33"]))
    # the same with shape blendshapes (model.py:88-92, 228-229): one beta for all frames, as TemporalSMPLify(shape=...) passes it
    beta = (2.0 * synth.uniform01(14, 9, 10) - 1.0).astype(np.float32)
    js, vs = bm.get_zero_pose_joint_and_vertex(t(beta.copy()).view(1, 10))
    grot_s, joint_s, vert_s = bm.forward_kinematics(t(pose.copy()), shape=t(beta.copy()).view(1, 10).expand(N, 10), tran=t(tran.copy()), calc_mesh=True)
    g["shape_beta"], g["shape_j0"], g["shape_v0_extra"] = beta, js[0].numpy(), vs[0][extra].numpy()
    g["shape_joint"], g["shape_vert_extra"] = joint_s.numpy(), vert_s[:, extra].numpy()
    g["shape_j33"] = torch.stack([sig_mp.sync_mp3d(vert_s[i], joint_s[i]) for i in range(N)]).numpy()
    # bbox normalisation (sig_mp.py:150-152 with get_bbox_scale L277-284)
    kp = (synth.uniform01(15, 0, N * 99)).reshape(N, 33, 3).astype(np.float32)
    kp[..., :2] = kp[..., :2] - 0.5
    outs = []
    for i in range(N):
        x = t(kp[i].copy())
        x[:, :2] = x[:, :2] / (sig_mp.get_bbox_scale(x))
        x[24:, :2] = x[24:, :2] - x[23:24, :2]
        x[:23, :2] = x[:23, :2] - x[23:24, :2]
        outs.append(x)
    g["bbox_in"], g["bbox_out"] = kp, torch.stack(outs).numpy()
    # lerp with a python-double weight (general.py:15-24)
    a = synth.normal(16, 0, 69)
    b = synth.normal(16, 1, 69)
    ks = np.array([0.0, 0.123456789, 0.5, 0.987654321, 1.0])
    g["lerp_a"], g["lerp_b"], g["lerp_k"] = a, b, ks
    g["lerp_out"] = np.stack([art.math.lerp(t(a), t(b), float(k)).numpy() for k in ks])
    # rotation matrix -> 6D (angular.py:267-274) and normalize_tensor with norms (general.py:27-39)
    g["rot2r6d_out"] = art.math.rotation_matrix_to_r6d(t(Rg.copy())).numpy()
    nx = synth.normal(19, 0, N * 7).reshape(N, 7).astype(np.float32)
    nx[3] = 0                                                 # zero row -> NaN, like the reference
    nn, nl = art.math.normalize_tensor(t(nx.copy()), return_norm=True)
    g["norm_in"], g["norm_out"], g["norm_len"] = nx, nn.numpy(), nl.numpy()
    # axis-angle -> R (angular.py:221-233), pure torch
    aa = (6 * synth.uniform01(17, 0, N * 3) - 3).reshape(N, 3).astype(np.float32)
    aa[0] = 0
    g["aa_in"] = aa
    g["aa_out"] = art.math.axis_angle_to_rotation_matrix(t(aa.copy())).numpy()
    # smplify forward residual: TemporalSMPLify.get_fitting_loss (temporal_smplify.py:198-220) ->
    # temporal_body_fitting_loss(output='reprojection') (losses.py:23-91). The object is built without its
    # __init__ (which needs the absent gmm_08.pkl); the prior is a zero callable -- it does not enter this output.
    from net.smplify import temporal_smplify as ts
    T = 24
    fit = object.__new__(ts.TemporalSMPLify)
    K = torch.tensor([[1450.0, 0.0, 960.0], [0.0, 1452.0, 540.0], [0.0, 0.0, 1.0]])
    fit.batch_size, fit.shape, fit.cam_k = T, None, K
    fit.ign_mp_joints = list(C.smplify_ignored_landmarks)
    fit.pose_prior = lambda p, _: torch.zeros(p.shape[0])
    fit.imu_ori = torch.eye(3).expand(T, 6, 3, 3).clone()
    rp = rand_rot(18, 0, T * 24).reshape(T, 24, 3, 3) * 0 + synth._rodrigues(
        0.4 * (synth.uniform01(18, 0, T * 72).reshape(T, 24, 3).astype(np.float64) - 0.5)).astype(np.float32)
    rt = np.float32([0.1, 0.2, 4.0]) + (0.5 * synth.uniform01(18, 1, T * 3).reshape(T, 3)).astype(np.float32)
    _, jj, vv = bm.forward_kinematics(t(rp.copy()), tran=t(rt.copy()), calc_mesh=True)
    j33 = ref_utils.sync_mp3d_from_smpl(vv, jj)
    proj = (K @ (j33 / j33[..., 2:]).unsqueeze(-1)).squeeze(-1)[..., :2]
    kp2 = torch.cat([proj + 40 * t(synth.normal(18, 2, T * 66).reshape(T, 33, 2)),
                     t(synth.uniform01(18, 3, T * 33).reshape(T, 33, 1).copy())], -1)
    kp2[3, :, :2] += 900.0                                   # one frame far off: robustifier saturates
    g["res_pose"], g["res_tran"], g["res_kp"], g["res_K"] = rp, rt, kp2.numpy().copy(), K.numpy()
    g["res_loss"] = fit.get_fitting_loss(t(rp.copy()), t(rt.copy()), kp2.clone()).numpy()
    g["res_gate_frame0_mean"] = np.float32(g["res_loss"].mean(-1)[0])
    _npz.save(os.path.join(OUT, "ops.npz"), **g)
    print("ops.npz:", {k: v.shape for k, v in g.items() if hasattr(v, "shape")})


# ------------------------------------------------------------------------------------------------------------
SCENARIOS = {
    # name: dict(T, conf kind or explicit schedule, first_tran, first_frame, flags)
    "mixed_firsttran": dict(T=128, conf="mixed", first_tran=True),
    "firstframe_high": dict(T=96, conf="mixed_hi0", first_frame=True),
    "firstframe_low": dict(T=64, conf="mixed_lo0", first_frame=True),
    "plain_lowstart": dict(T=96, conf="lowstart"),
    "teleport": dict(T=48, conf="high", first_tran=(20.0, 1.0, 6.0)),
    "noflat_mid": dict(T=96, conf="midheavy", first_tran=True, use_flat_floor=False),
    "live_post": dict(T=96, conf="mixed", first_frame=True, live="post"),
    "live_pre": dict(T=96, conf="livepre", first_frame=True, live="pre"),
    "reproj_opt": dict(T=96, conf="mixed_hi0", first_tran=True, use_reproj_opt=True),
    "long_mixed": dict(T=512, conf="mixed", first_tran=True),       # the north-star sequence length
    "no_updaters": dict(T=72, conf="lowstart", first_frame=True, use_vision_updater=False, use_imu_updater=False),
    # every frame visible at the north-star length: the regime the sequence-mode engine (hoisted input projections,
    # skewed stage pipeline) accelerates; mid stretches inside exercise the lerp there
    "allvis_long": dict(T=512, conf="allvis", first_tran=True),
    "allvis_ff": dict(T=160, conf="allvis", first_frame=True),
}


def _conf(kind, T, seed):
    if kind in ("mixed", "high", "mid", "low", "occ"):
        return synth.conf_schedule(seed, 7, T, kind)
    u = synth.uniform01(seed, 70, T).astype(np.float64)
    c = synth.conf_schedule(seed, 7, T, "mixed")
    if kind == "mixed_hi0":
        c[:20] = 0.86 + 0.1 * u[:20]
    elif kind == "mixed_lo0":
        c[:12] = 0.4 + 0.25 * u[:12]
        c[12:40] = 0.85 + 0.1 * u[12:40]
    elif kind == "lowstart":
        c[:25] = 0.35 + 0.3 * u[:25]
        c[25:35] = 0.72 + 0.06 * u[25:35]
        c[35:80] = 0.84 + 0.12 * u[35:80]
    elif kind == "midheavy":
        c[:] = np.where(u < 0.6, 0.715 + 0.07 * u / 0.6, c)
    elif kind == "allvis":           # c > 0.7 everywhere: high with mid stretches
        c = synth.conf_schedule(seed, 7, T, "high")
        seg = (np.arange(T) // 37) % 4
        c = np.where(seg == 2, 0.715 + 0.07 * u, c)
        c[0] = 0.9
    elif kind == "livepre":          # thresholds (0.85, 0.9)
        seg = (np.arange(T) // 16) % 3
        c = np.where(seg == 0, 0.915 + 0.07 * u, np.where(seg == 1, 0.862 + 0.026 * u, 0.5 + 0.3 * u))
    return c


def run_reference(sig_mp, sd, inputs, sc):
    """One scenario through the REFERENCE's ``Net.forward_online`` frame loop on the given inputs (j2dc [T,33,3],
    accc [T,6,3], oric [T,6,3,3], gravityc [3], first_tran [3] or empty). Returns the dict of everything a fixture stores
    besides its inputs: pose, tran, trace, net_out (last linear2 output of every sub-net per frame), final (h, c)."""
    Net = sig_mp.Net
    T = inputs["j2dc"].shape[0]
    live = sc.get("live")
    Net.live = (live == "pre")
    Net.update_vision_count = 0
    Net.j_temp = None
    net = Net()
    net.load_state_dict(sd, strict=True)
    net.eval()
    if live == "post":
        net.live = True
    net.use_flat_floor = bool(sc.get("use_flat_floor", True))
    net.use_reproj_opt = bool(sc.get("use_reproj_opt", False))
    net.use_vision_updater = bool(sc.get("use_vision_updater", True))
    net.use_imu_updater = bool(sc.get("use_imu_updater", True))
    net.gravityc = torch.from_numpy(np.asarray(inputs["gravityc"], np.float32).copy())
    ft = inputs.get("first_tran")
    ft = None if ft is None or np.size(ft) == 0 else torch.tensor(np.asarray(ft, np.float32))
    ff = bool(sc.get("first_frame", False))
    calls = []                                               # hooks: every linear2 call, in order
    hooks = [getattr(net, n).linear2.register_forward_hook(
        lambda mod, inp, out, n=n: calls.append((n, out.detach().numpy().reshape(-1).copy())))
        for n, *_ in C.NETS]
    poses, trans, trace, outs = [], [], [], []
    for t in range(T):
        calls.clear()
        j2 = torch.from_numpy(inputs["j2dc"][t].copy())
        ac = torch.from_numpy(inputs["accc"][t].copy())
        ori = torch.from_numpy(inputs["oric"][t].copy())
        n_floor0 = len(net.floor_y)
        reach0 = net.first_reach
        if t == 0:
            p, tr = net.forward_online(j2, ac, ori, first_tran=ft, first_frame=ff)
        else:
            p, tr = net.forward_online(j2, ac, ori)
        poses.append(p.numpy().copy())
        trans.append(tr.numpy().copy())
        order = [n for n, _ in calls]
        rec = np.zeros(6 * 144 + 16, np.float32)             # last output of each net this frame (padded)
        for n, o in calls:
            k = C.NET_INDEX[n]
            rec[k * 144:k * 144 + o.size] = o
        outs.append(rec)
        trace.append([float(inputs["j2dc"][t, :, 2].mean()), order.count("rnn4"), order.count("rnn6"),
                      len(net.floor_y) - n_floor0, len(net.floor_y), int(reach0 and not net.first_reach),
                      int(net.update_vision_count) if live else 0])
    for h in hooks:
        h.remove()
    res = dict(pose=np.stack(poses), tran=np.stack(trans), trace=np.asarray(trace, np.float64), net_out=np.stack(outs),
               last_pfoot=net.last_pfoot.numpy())
    for n, *_ in C.NETS:
        slot = int(n[3:]) - 1
        hc = net.hidden[slot]
        res["h_" + n] = hc[0].numpy()[:, 0].copy()
        res["c_" + n] = hc[1].numpy()[:, 0].copy()
    Net.live = False
    return res


def _scenario_of(z):
    """flags stored in a fixture -> the scenario dict run_reference takes"""
    return dict(live=str(z["live"]) or None, first_frame=bool(int(z["first_frame"])), use_flat_floor=bool(int(z["use_flat_floor"])),
                use_reproj_opt=bool(int(z["use_reproj_opt"])), use_vision_updater=bool(int(z["use_vision_updater"])),
                use_imu_updater=bool(int(z["use_imu_updater"])))


def _sha256(path):
    import hashlib
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def write_hashes():
    """sha256 of every tests/golden/*.npz into meta.json (tests/test_oracle_golden.py verifies them on every run)."""
    import json
    path = os.path.join(OUT, "meta.json")
    with open(path) as f:
        meta = json.load(f)
    meta["sha256"] = {n: _sha256(os.path.join(OUT, n)) for n in sorted(os.listdir(OUT)) if n.endswith(".npz")}
    with open(path, "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


def capture_sequences(art, sig_mp, body):
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(WEIGHT_SEED).items()}
    meta = {"weight_seed": WEIGHT_SEED, "body_seed": BODY_SEED,
            "weight_checksum": {k: synth.checksum(v.numpy()) for k, v in list(sd.items())[::7]},
            "body_checksum": {k: synth.checksum(v) for k, v in body.items()},
            "torch": torch.__version__, "threads": torch.get_num_threads()}
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    for si, (name, sc) in enumerate(SCENARIOS.items()):
        if only and name not in only:
            continue
        T = sc["T"]
        mseed = 100 + si
        conf = _conf(sc["conf"], T, mseed * 7919)
        m = synth.make_motion(mseed, 1, T, body, conf=conf)
        ft = sc.get("first_tran")
        if ft is True:
            ft = m["first_tran"][0]
        ft = np.zeros(0, np.float32) if ft is None else np.asarray(ft, np.float32)
        inputs = dict(j2dc=m["j2dc"][0], accc=m["accc"][0], oric=m["oric"][0], gravityc=m["gravityc"][0], first_tran=ft)
        res = run_reference(sig_mp, sd, inputs, sc)
        _npz.save(
            os.path.join(OUT, f"seq_{name}.npz"), **inputs, first_frame=np.int32(bool(sc.get("first_frame", False))),
            live=np.str_(sc.get("live") or ""), use_flat_floor=np.int32(sc.get("use_flat_floor", True)),
            use_reproj_opt=np.int32(sc.get("use_reproj_opt", False)),
            use_vision_updater=np.int32(sc.get("use_vision_updater", True)), use_imu_updater=np.int32(sc.get("use_imu_updater", True)),
            **res)
        tr = res["trace"]
        print(f"seq_{name}: T={T} c>=hi {np.mean(tr[:,0]>=0.8):.2f} rnn4x2 {int((tr[:,1]==2).sum())} "
              f"rnn6x2 {int((tr[:,2]==2).sum())} floor {int(tr[-1,4])} reach@{np.argmax(tr[:,5]) if tr[:,5].any() else -1} "
              f"|tran| {np.abs(res['tran']).max():.2f}")
    import json
    if not only:
        with open(os.path.join(OUT, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
    write_hashes()


def check_sequences(sig_mp):
    """--check: re-run the reference on the STORED inputs of every committed sequence fixture and require the stored
    outputs back bit for bit (pose, tran, branch trace, sub-net outputs, final states). Independent of today's
    ``synth.make_motion``: it proves the fixtures are the reference's own numbers for the inputs they carry."""
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(WEIGHT_SEED).items()}
    bad = 0
    for name in sorted(os.listdir(OUT)):
        if not (name.startswith("seq_") and name.endswith(".npz")):
            continue
        z = np.load(os.path.join(OUT, name))
        inputs = {k: z[k] for k in ("j2dc", "accc", "oric", "gravityc", "first_tran")}
        res = run_reference(sig_mp, sd, inputs, _scenario_of(z))
        diffs = {k: (float(np.abs(v.astype(np.float64) - z[k].astype(np.float64)).max()) if v.size else 0.0) for k, v in res.items()}
        ok = all(np.array_equal(v, z[k]) for k, v in res.items())
        bad += 0 if ok else 1
        print(f"{name}: {'bit-exact' if ok else 'MISMATCH ' + str({k: d for k, d in diffs.items() if d})}  (T={inputs['j2dc'].shape[0]})")
    import json
    with open(os.path.join(OUT, "meta.json")) as f:
        want = json.load(f).get("sha256", {})
    for n, h in want.items():
        if _sha256(os.path.join(OUT, n)) != h:
            bad += 1
            print(f"{n}: sha256 differs from meta.json")
    print("check:", "OK" if bad == 0 else f"{bad} problem(s)")
    return bad


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    art, sig_mp, body = import_reference()
    torch.manual_seed(0)
    with torch.no_grad():
        if "--check" in sys.argv[1:]:
            sys.exit(1 if check_sequences(sig_mp) else 0)
        if not [a for a in sys.argv[1:] if not a.startswith("-")]:
            capture_ops(art, sig_mp, body)
        capture_sequences(art, sig_mp, body)
