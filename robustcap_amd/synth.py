"""Seeded synthetic assets for the sig_mp hot path: weights, an SMPL-format body and 60 fps motion inputs.

The reference ships none of its assets (weights, SMPL pickle, datasets are external downloads, SURVEY.md
section 0 fact 9), so parity and benchmarks run on assets regenerated from integer seeds by THIS module, on
any machine, bit-identically (counter-based splitmix64 -> 24-bit uniforms, numpy only).

  * make_state_dict(seed)  -- tensors with the key names / shapes of the reference ``Net.state_dict()``
                              (SURVEY.md A.2); torch-style uniform(-1/sqrt(fan), 1/sqrt(fan)) plus output biases
                              that put the nets in their physical operating range (see below).
  * make_body(seed)        -- arrays with the fields articulate/model.py:31-38 reads from the SMPL pickle.
  * make_motion(...)       -- per-body camera-frame inputs (33 keypoints x (x/z, y/z, conf), 6 IMU
                              accelerations and orientations) following the reference's own IMU synthesis
                              recipe (preprocess.py:22-33, 220-222) and SURVEY.md section 8(d) config 2/4.
"""
import numpy as np

from . import config as C

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform01(seed, stream, n):
    """n float32 uniforms in [0,1): element i = top 24 bits of splitmix64(splitmix64(seed, stream) + i)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))
        idx = np.arange(n, dtype=np.uint64)
        z = _splitmix64(base + idx)
    return ((z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def normal(seed, stream, n):
    """float32 standard normals (Box-Muller on two uniform streams)."""
    u1 = uniform01(seed, stream * 2 + 1000003, n).astype(np.float64)
    u2 = uniform01(seed, stream * 2 + 1000004, n).astype(np.float64)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    return (r * np.cos(2.0 * np.pi * u2)).astype(np.float32)


# ----------------------------------------------------------------------------------------------------- weights
def make_state_dict(seed=0, gain=1.0):
    """Ordered {key: float32 ndarray} with the reference state_dict layout.

    Every tensor is uniform(-k, k) * gain with torch's default k (LSTM: 1/sqrt(H); Linear: 1/sqrt(fan_in)).
    Four output biases are then shifted so that a random-weight network still lives where the trained one
    does -- otherwise the per-frame logic degenerates (root depth ~0 makes the re-projection x/z blow up):
      rnn6.linear2.bias += tran_offset (0, 0.25, 5)      -> camera-frame root position a few metres away
      rnn7.linear2.bias += identity 6D (1,0,0,0,1,0)x24  -> well-conditioned Gram-Schmidt
      rnn8.linear2: weight * 12, bias += (0.8, 0.5)      -> foot-contact probabilities that cross 0.7 both ways
      rnn3.linear2: weight * 4                           -> root velocities of a few cm/frame
    """
    sd = {}
    for stream, (key, shape) in enumerate(C.state_dict_spec()):
        n = int(np.prod(shape))
        if ".rnn." in key:
            k = 1.0 / np.sqrt(shape[0] // 4)
        elif key.endswith(".weight"):
            k = 1.0 / np.sqrt(shape[1])
        else:  # Linear bias: fan_in of its weight
            wshape = dict(C.state_dict_spec())[key[:-4] + "weight"]
            k = 1.0 / np.sqrt(wshape[1])
        u = uniform01(seed, stream, n)
        sd[key] = ((u * np.float32(2.0) - np.float32(1.0)) * np.float32(k * gain)).reshape(shape).astype(np.float32)
    sd["rnn6.linear2.bias"] = (sd["rnn6.linear2.bias"] + np.asarray(C.tran_offset, np.float32)).astype(np.float32)
    sd["rnn7.linear2.bias"] = (sd["rnn7.linear2.bias"] +
                               np.tile(np.asarray([1, 0, 0, 0, 1, 0], np.float32), 24)).astype(np.float32)
    sd["rnn8.linear2.weight"] = (sd["rnn8.linear2.weight"] * np.float32(12.0)).astype(np.float32)
    sd["rnn8.linear2.bias"] = (sd["rnn8.linear2.bias"] + np.asarray([0.8, 0.5], np.float32)).astype(np.float32)
    sd["rnn3.linear2.weight"] = (sd["rnn3.linear2.weight"] * np.float32(4.0)).astype(np.float32)
    return sd


def checksum(a):
    """Order-sensitive 64-bit checksum of an array's raw bytes (fixtures record these instead of the data)."""
    b = np.ascontiguousarray(a).reshape(-1).view(np.uint8).astype(np.uint64)
    with np.errstate(over="ignore"):
        w = (np.arange(b.size, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(1)) & _M64
        return int((b * w).sum(dtype=np.uint64) & _M64)


# -------------------------------------------------------------------------------------------------------- body
_REST = np.array([
    [0.00, 0.00, 0.00], [0.07, -0.09, -0.01], [-0.07, -0.09, -0.01], [0.00, 0.11, -0.03],
    [0.10, -0.47, -0.01], [-0.10, -0.47, -0.01], [0.00, 0.25, 0.00], [0.09, -0.87, -0.04],
    [-0.09, -0.87, -0.04], [0.00, 0.30, 0.03], [0.11, -0.93, 0.08], [-0.11, -0.93, 0.08],
    [0.00, 0.52, -0.02], [0.08, 0.42, -0.01], [-0.08, 0.42, -0.01], [0.00, 0.60, 0.03],
    [0.19, 0.44, -0.02], [-0.19, 0.44, -0.02], [0.45, 0.43, -0.04], [-0.45, 0.43, -0.04],
    [0.70, 0.44, -0.04], [-0.70, 0.44, -0.04], [0.79, 0.43, -0.05], [-0.79, 0.43, -0.05]], np.float64)


def make_body(seed=1, num_vertex=6890):
    """Synthetic SMPL-format body: dict(J[24,3], v_template[V,3], weights[V,24], parent[24]) float32.

    A hand-authored T-pose skeleton (+ seeded jitter) with vertices scattered round the bones and sparse
    skinning weights (<= 4 joints per vertex, rows sum to 1). Only the layout matters for the hot path:
    ``forward_kinematics`` reads J, v_template, weights and the parent table (articulate/model.py:31-39, 209-241).
    """
    V = num_vertex
    parent = np.asarray(C.smpl_parent, np.int64)
    J = _REST + np.array([0.0, -0.23, 0.02]) + 0.005 * normal(seed, 1, 72).reshape(24, 3)
    J = J.astype(np.float32)
    bone = 1 + (uniform01(seed, 2, V) * 23).astype(np.int64).clip(0, 22)           # vertex sits on bone p->b
    t = uniform01(seed, 3, V)[:, None]
    p = parent[bone]
    v = J[p] * (1 - t) + J[bone] * t + 0.04 * normal(seed, 4, V * 3).reshape(V, 3)
    w = np.zeros((V, 24), np.float32)
    a = 0.55 + 0.4 * uniform01(seed, 5, V)
    r = uniform01(seed, 6, V * 3).reshape(V, 3)
    gp = np.where(parent[p] < 0, p, parent[p])
    other = (uniform01(seed, 7, V) * 24).astype(np.int64).clip(0, 23)
    rest = (1 - a)[:, None] * r / r.sum(1, keepdims=True)
    idx = np.arange(V)
    np.add.at(w, (idx, p), a.astype(np.float32))
    np.add.at(w, (idx, bone), rest[:, 0].astype(np.float32))
    np.add.at(w, (idx, gp), rest[:, 1].astype(np.float32))
    np.add.at(w, (idx, other), rest[:, 2].astype(np.float32))
    w = (w / w.sum(1, keepdims=True)).astype(np.float32)
    # shape blendshapes and joint regressor (articulate/model.py:33-35, used only with shape=...): seeded directions of a
    # few centimetres per unit beta; every regressor row is a convex combination of 24 seeded vertices. (Like the official
    # model, J is what the model file stores; with shape=None the regressor is never applied, model.py:86.)
    sd = (0.015 * normal(seed, 8, V * 30)).reshape(V, 3, 10).astype(np.float32)
    Jr = np.zeros((24, V), np.float32)
    for j in range(24):
        ids = (uniform01(seed, 20 + j, 24).astype(np.float64) * V).astype(np.int64) % V
        ww = uniform01(seed, 60 + j, 24).astype(np.float64) + 0.1
        np.add.at(Jr[j], ids, (ww / ww.sum()).astype(np.float32))
    return {"J": J, "v_template": v.astype(np.float32), "weights": w, "parent": parent, "shapedirs": sd, "J_regressor": Jr}


# ------------------------------------------------------------------------------------------------------ motion
def _rodrigues(aa):
    """[...,3] axis-angle -> [...,3,3] (float64)."""
    th = np.linalg.norm(aa, axis=-1, keepdims=True)
    k = aa / np.where(th < 1e-12, 1.0, th)
    K = np.zeros(aa.shape[:-1] + (3, 3))
    K[..., 0, 1], K[..., 0, 2] = -k[..., 2], k[..., 1]
    K[..., 1, 0], K[..., 1, 2] = k[..., 2], -k[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -k[..., 1], k[..., 0]
    s, c = np.sin(th)[..., None], np.cos(th)[..., None]
    return np.eye(3) + s * K + (1 - c) * (K @ K)


def _smooth(x, n):
    """moving average of window n along axis 0 (edge-padded), keeps length; works for any length >= 1."""
    if n <= 1 or x.shape[0] < 2:
        return x
    lo = n // 2
    pad = np.concatenate([np.repeat(x[:1], lo, 0), x, np.repeat(x[-1:], n - 1 - lo, 0)], 0)
    c = np.cumsum(np.concatenate([np.zeros_like(pad[:1]), pad], 0), 0)
    return (c[n:] - c[:-n]) / n


def body_fk_numpy(body, pose, tran, vertex_ids):
    """float64 FK + skinning of selected vertices. pose [N,24,3,3] local, tran [N,3] ->
    (global rotations [N,24,3,3], joints [N,24,3], vertices [N,len(ids),3])."""
    parent = body["parent"]
    J = body["J"].astype(np.float64)
    J = J - J[:1]
    vt = body["v_template"].astype(np.float64)[list(vertex_ids)] - body["J"].astype(np.float64)[:1]
    w = body["weights"].astype(np.float64)[list(vertex_ids)]
    N = pose.shape[0]
    G = np.zeros((N, 24, 3, 3))
    P = np.zeros((N, 24, 3))
    G[:, 0] = pose[:, 0]
    for i in range(1, 24):
        G[:, i] = G[:, parent[i]] @ pose[:, i]
        P[:, i] = P[:, parent[i]] + np.einsum("nij,j->ni", G[:, parent[i]], J[i] - J[parent[i]])
    # skinning: v' = sum_j w_j (G_j (v - J_j) + P_j)
    d = vt[None, :, None, :] - J[None, None, :, :]                          # [1,V,24,3]
    vj = np.einsum("njab,zvjb->nvja", G, d) + P[:, None, :, :]              # [N,V,24,3]
    v = np.einsum("vj,nvja->nva", w, vj)
    return G, P + tran[:, None, :], v + tran[:, None, :]


def conf_schedule(seed, stream, T, kind):
    """per-frame mean keypoint confidence, kept >= 0.012 away from the 0.7 / 0.8 thresholds.

    kind: 'high' (all frames >= 0.8), 'mid', 'low', 'mixed' (50 % high / 20 % mid / 30 % low in runs of 30-120
    frames) or 'occ' (config 4: high with occluded runs that drop below 0.7)."""
    u = uniform01(seed, stream, 4 * T + 16)
    c = np.empty(T, np.float64)
    t, q = 0, 0
    while t < T:
        run = 30 + int(u[q] * 91)
        r = u[q + 1]
        q += 2
        if kind == "mixed":
            reg = "high" if r < 0.5 else ("mid" if r < 0.7 else "low")
        elif kind == "occ":
            reg = "high" if r < 0.6 else "low"
        else:
            reg = kind
        lo, hi = {"high": (0.83, 0.97), "mid": (0.715, 0.785), "low": (0.30, 0.68)}[reg]
        n = min(run, T - t)
        seg = lo + (hi - lo) * u[q:q + 1] + 0.0 * np.arange(n)
        drift = (hi - lo) * 0.15 * np.sin(np.arange(n) / 17.0 + 6.28 * u[q + 1])
        c[t:t + n] = np.clip(seg + drift, lo, hi)
        q += 2
        t += n
    return c


def motion_trajectory(s, T):
    """One body's seeded trajectory (float64): local rotations R [T,24,3,3] with the root expressed in the camera frame, the gravity
    direction g [3] in that frame and the root translation tr [T,3] (a smooth walk inside x,y in [-1,1], z in [3,8])."""
    # smooth local axis-angle trajectories, amplitude ~0.5 rad
    aa = np.cumsum(0.02 * normal(s, 1, T * 72).reshape(T, 24, 3).astype(np.float64), 0)
    aa = _smooth(aa, 9)
    aa = 0.7 * np.tanh(aa / 0.7)
    R = _rodrigues(aa)
    # root: camera looks along +z, image y points down -> body up (+y) maps to -y; slow yaw + small tilt
    yaw = _smooth(np.cumsum(0.03 * normal(s, 2, T).astype(np.float64)), 15) + 6.28 * uniform01(s, 3, 1)[0]
    tilt = 0.15 * (uniform01(s, 4, 3).astype(np.float64) - 0.5)
    Rtilt = _rodrigues(tilt)
    flip = np.diag([1.0, -1.0, -1.0])
    Ry = _rodrigues(np.stack([np.zeros(T), yaw, np.zeros(T)], -1))
    R[:, 0] = Rtilt @ flip @ Ry @ R[:, 0]
    g = Rtilt @ flip @ np.array([0.0, -1.0, 0.0])
    u0 = uniform01(s, 5, 3).astype(np.float64)
    start = np.array([-0.8 + 1.6 * u0[0], -0.5 + 1.0 * u0[1], 3.5 + 3.5 * u0[2]])
    walk = _smooth(np.cumsum(0.012 * normal(s, 6, T * 3).reshape(T, 3).astype(np.float64), 0), 15)
    tr = start + 1.2 * np.tanh(walk / 1.2)
    return R, g, tr


def motion_confidence(s, T, conf):
    """Per-keypoint confidences ck [T,33] of one body (schedule + per-keypoint spread) and the unit keypoint noise [T,33,2]."""
    c = conf_schedule(s, 7, T, conf) if isinstance(conf, str) else np.asarray(conf, np.float64)
    dk = 0.04 * (uniform01(s, 8, T * 33).reshape(T, 33).astype(np.float64) - 0.5)
    dk -= dk.mean(1, keepdims=True)
    return np.clip(c[:, None] + dk, 0.0, 1.0), normal(s, 9, T * 66).reshape(T, 33, 2)


def make_motion(seed, B, T, body, conf="mixed", noise=0.003):
    """Synthetic 60 fps camera-frame sequences for B bodies x T frames (all float32):

    j2dc [B,T,33,3]  (x/z, y/z, confidence)      accc [B,T,6,3]      oric [B,T,6,3,3]
    gravityc [B,3]   first_tran [B,3]            plus ground truth pose [B,T,24,3,3], tran [B,T,3].
    IMU recipe: ori = global rotation of joints ji_mask, acc = smoothed second difference * 3600 of vertices
    vi_mask (preprocess.py:22-33, 221-222), both expressed in the camera frame.
    (robustcap_amd.preprocess.make_motion_device: the same sequences with FK, IMU synthesis and projection on the GPU.)
    """
    ids = list(C.mp_mask) + list(C.vi_mask)
    out = {k: [] for k in ("j2dc", "accc", "oric", "gravityc", "first_tran", "pose", "tran", "conf")}
    for b in range(B):
        s = seed * 7919 + b
        R, g, tr = motion_trajectory(s, T)
        G, joint, vert = body_fk_numpy(body, R, tr, ids)
        v33, v6 = vert[:, :33].copy(), vert[:, 33:]
        for row, j in C.mp_joint_override.items():
            v33[:, row] = joint[:, j]
        ori = G[:, list(C.ji_mask)]
        acc = np.zeros((T, 6, 3))
        if T > 2:
            acc[1:-1] = (v6[:-2] + v6[2:] - 2 * v6[1:-1]) * 3600
        if T > 4:
            acc[2:-2] = (v6[:-4] + v6[4:] - 2 * v6[2:-2]) * 3600 / 4
        ck, unit = motion_confidence(s, T, conf)
        uv = v33[..., :2] / v33[..., 2:]
        uv = uv + noise * (1 - ck)[..., None] * unit
        out["j2dc"].append(np.concatenate([uv, ck[..., None]], -1))
        out["accc"].append(acc)
        out["oric"].append(ori)
        out["gravityc"].append(g)
        out["first_tran"].append(tr[0])
        out["pose"].append(R)
        out["tran"].append(tr)
        out["conf"].append(ck.mean(1))
    return {k: np.stack(v).astype(np.float32) for k, v in out.items()}


# ----------------------------------------------------------------------------------------------------- dataset
def _log_map(R):
    """[...,3,3] rotation matrices -> axis-angle [...,3] (float64, angle < pi)."""
    v = np.stack([R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] - R[..., 0, 1]], -1) * 0.5
    s = np.linalg.norm(v, axis=-1)
    c = (np.trace(R, axis1=-2, axis2=-1) - 1) * 0.5
    th = np.arctan2(s, c)
    k = np.where(s > 1e-12, th / np.where(s > 1e-12, s, 1.0), 1.0)
    return v * k[..., None]


def make_dataset(seed, n_seq, T, body, n_cam=3, conf="mixed", image_size=(1920, 1080)):
    """Synthetic multi-camera dataset in the layout of the reference's preprocessed ``test.pt``
    (preprocess.py:229-237, read by evaluate.py:24-52): per sequence world-frame ``pose`` [T,72] axis-angle,
    ``tran`` [T,3], ``imu_ori`` [T,6,3,3], ``imu_acc`` [T,6,3]; per camera ``cam_K`` [n_cam,3,3], ``cam_T`` [n_cam,4,4]
    (T_cw) and ``joint2d_mp`` [n_cam,T,33,3] = (u / width, v / height, confidence)."""
    W, H = image_size
    ds = {k: [] for k in ("name", "pose", "tran", "imu_ori", "imu_acc", "cam_K", "cam_T", "joint2d_mp")}
    ids = list(C.mp_mask)
    for i in range(n_seq):
        m = make_motion(seed * 131 + i, 1, T, body, conf=conf)          # camera-0 frame == world frame here
        R = m["pose"][0].astype(np.float64)
        tr = m["tran"][0].astype(np.float64)
        G, joint, vert = body_fk_numpy(body, R, tr, ids)
        v33 = vert.copy()
        for row, j in C.mp_joint_override.items():
            v33[:, row] = joint[:, j]
        Ks, Ts, kps = [], [], []
        for c in range(n_cam):
            u = uniform01(seed * 977 + i, 50 + c, 8).astype(np.float64)
            yaw = 0.0 if c == 0 else 0.5 * (u[0] - 0.5)
            Rcw = np.eye(3) if c == 0 else _rodrigues(np.array([0.05 * (u[1] - 0.5), yaw, 0.03 * (u[2] - 0.5)]))
            centre = tr.mean(0)
            t = np.zeros(3) if c == 0 else centre - Rcw @ centre + np.array([0.4 * (u[3] - 0.5), 0.2 * (u[4] - 0.5), 0.6 * u[5]])
            Tcw = np.eye(4)
            Tcw[:3, :3], Tcw[:3, 3] = Rcw, t
            f = 1400.0 + 200.0 * u[6]
            K = np.array([[f, 0.0, W / 2 + 20 * (u[7] - 0.5)], [0.0, f * 1.002, H / 2], [0.0, 0.0, 1.0]])
            Xc = v33 @ Rcw.T + t
            uv = (Xc / Xc[..., 2:]) @ K.T
            conf_c = m["j2dc"][0][..., 2].astype(np.float64)
            noise = 2.0 * (1 - conf_c)[..., None] * normal(seed * 31 + i, 80 + c, T * 66).reshape(T, 33, 2)
            kp = np.concatenate([(uv[..., :2] + noise) / np.array([W, H]), conf_c[..., None]], -1)
            Ks.append(K), Ts.append(Tcw), kps.append(kp)
        ds["name"].append(f"synth_seq{i:03d}_cAll")
        ds["pose"].append(_log_map(R).reshape(T, 72).astype(np.float32))
        ds["tran"].append(tr.astype(np.float32))
        ds["imu_ori"].append(m["oric"][0])
        ds["imu_acc"].append(m["accc"][0])
        ds["cam_K"].append(np.stack(Ks).astype(np.float32))
        ds["cam_T"].append(np.stack(Ts).astype(np.float32))
        ds["joint2d_mp"].append(np.stack(kps).astype(np.float32))
    return ds


# ------------------------------------------------------------------------------------------------- smplify prior
def make_gmm(seed=3, n=8, dim=69):
    """Synthetic stand-in for the SMPLify pose prior ``gmm_08.pkl`` (external asset; net/smplify/prior.py:102-111 reads
    a dict with 'means' [n,dim], 'covars' [n,dim,dim], 'weights' [n]): seeded means, low-rank + diagonal SPD covariances."""
    means = 0.2 * normal(seed, 1, n * dim).reshape(n, dim).astype(np.float64)
    Bm = 0.15 * normal(seed, 2, n * dim * 6).reshape(n, dim, 6).astype(np.float64)
    covars = Bm @ np.swapaxes(Bm, 1, 2) + (0.08 ** 2) * np.eye(dim)[None]
    w = uniform01(seed, 3, n).astype(np.float64) + 0.5
    return {"means": means, "covars": covars, "weights": w / w.sum()}


# ------------------------------------------------------------------------------------------- metric regressor
def make_j_regressor(seed=4, n_joint=17, num_vertex=6890, support=48):
    """Synthetic stand-in for ``J_regressor_h36m.npy`` (external asset, evaluate.py:17): [17, V] float32, every row a
    convex combination (non-negative, sums to 1) of ``support`` seeded vertices -- the structure of the real regressor."""
    Jr = np.zeros((n_joint, num_vertex), np.float32)
    for k in range(n_joint):
        ids = (uniform01(seed, 2 * k, support).astype(np.float64) * num_vertex).astype(np.int64) % num_vertex
        w = uniform01(seed, 2 * k + 1, support).astype(np.float64) + 0.05
        np.add.at(Jr[k], ids, (w / w.sum()).astype(np.float32))
    return Jr
