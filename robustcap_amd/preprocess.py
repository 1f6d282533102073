"""IMU synthesis of the reference's dataset preparation on the GPU.

Mirrors ``preprocess.py``: ``_syn_acc`` (L22-33) and the recipe of L206-214 that turns SMPL pose + translation into the
six virtual IMU readings (global orientation of joints ``ji_mask``, acceleration of vertices ``vi_mask``). The
arithmetic runs in librobustcap_hip.so (rc_syn_acc, rc_synth_imu).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from . import body as _body
from . import config as cfg


def _syn_acc(v, smooth_n=2, device="cuda"):
    """preprocess._syn_acc: v [T, ...] positions (60 fps) -> accelerations of the same shape (device tensor)."""
    x = _body._f32c(v, torch.device(device))
    T = x.shape[0]
    width = int(np.prod(x.shape[1:])) if x.dim() > 1 else 1
    out = torch.empty_like(x)
    lib = _lib.load()
    rc = lib.rc_syn_acc(_lib.ptr(x), T, width, int(smooth_n), _lib.ptr(out), _lib.stream_ptr())
    if rc != 0:
        raise _lib.RobustcapLibraryError(f"rc_syn_acc failed ({rc}): needs smooth_n >= 1 and, for smooth_n >= 2, "
                                         f"at least 2 * smooth_n + 1 frames (got T={T}, smooth_n={smooth_n})")
    return out


def synthesize_imu(model, pose, tran, smooth_n=2, vertex_ids=cfg.vi_mask, joint_ids=cfg.ji_mask):
    """preprocess.py:206-214 for one sequence. ``model`` = robustcap_amd.body.ParametricModel, pose [T,24,3,3] local
    rotation matrices (or [T,72] axis-angle), tran [T,3]. Returns device tensors
    (imu_ori [T,6,3,3], imu_acc [T,6,3], joint3d [T,24,3], vert6 [T,6,3])."""
    dev = model.device
    pose = torch.as_tensor(pose)
    if pose.shape[-1] == 72 or (pose.dim() == 3 and pose.shape[-2:] == (24, 3)):
        pose = _body.axis_angle_to_rotation_matrix(pose.reshape(-1, 3), dev)
    pose = _body._f32c(pose, dev).view(-1, 24, 3, 3)
    T = pose.shape[0]
    tran = _body._f32c(tran, dev).view(T, 3)
    model._ensure_mesh()
    ori, acc = torch.empty(T, 6, 3, 3, device=dev), torch.empty(T, 6, 3, device=dev)
    joint, vert6 = torch.empty(T, 24, 3, device=dev), torch.empty(T, 6, 3, device=dev)
    vid = (C.c_int32 * 6)(*[int(v) for v in vertex_ids])
    jid = (C.c_int32 * 6)(*[int(j) for j in joint_ids])
    rc = model._lib.rc_synth_imu(model._ctx, _lib.ptr(pose), _lib.ptr(tran), vid, jid, T, int(smooth_n), _lib.ptr(ori), _lib.ptr(acc),
                                 _lib.ptr(joint), _lib.ptr(vert6), _lib.stream_ptr())
    _lib.check(model._ctx, rc, "rc_synth_imu")
    return ori, acc, joint, vert6


def make_motion_device(seed, B, T, body, conf="mixed", noise=0.003, model=None, device="cuda"):
    """``synth.make_motion`` with the heavy part on the GPU ("on-device generator for benchmark inputs (FK -> ori /
    acc / 2D), removes host prep from large-B runs"; the recipe is preprocess.py:22-33, 206-222 + the projection of evaluate.py:70-72).

    The seeded random walks (24 axis-angles, root yaw / tilt, root translation, confidence schedule, unit keypoint noise) are a few
    hundred floats per frame and stay host numpy -- the SAME numbers as ``make_motion``; forward kinematics + the 33 landmarks
    (``rc_body_fk``), the six virtual IMUs (``rc_synth_imu``: global orientation of joints ji_mask, smoothed second difference of vertices
    vi_mask) and the projection run on the device and the outputs stay there. Returns the dict of ``make_motion`` with device tensors for
    j2dc / accc / oric / pose / tran and host arrays for gravityc / first_tran / conf; equal to the host generator to fp32 rounding."""
    from . import synth
    dev = torch.device(device)
    model = model or _body.ParametricModel(body=body, device=device)
    R, tr, ck, unit, grav = [], [], [], [], []
    for b in range(B):
        s = seed * 7919 + b
        r, g, t_ = synth.motion_trajectory(s, T)
        c, u = synth.motion_confidence(s, T, conf)
        R.append(r); tr.append(t_); ck.append(c); unit.append(u); grav.append(g)
    pose = torch.from_numpy(np.stack(R).astype(np.float32)).to(dev)                      # [B,T,24,3,3]
    tran = torch.from_numpy(np.stack(tr).astype(np.float32)).to(dev)                      # [B,T,3]
    ckd = torch.from_numpy(np.stack(ck).astype(np.float32)).to(dev)                       # [B,T,33]
    unitd = torch.from_numpy(np.stack(unit).astype(np.float32)).to(dev)                   # [B,T,33,2]
    _, _, j33 = model.forward_kinematics(pose.view(-1, 24, 3, 3), tran=tran.view(-1, 3), calc_mesh=True)
    j33 = j33.view(B, T, 33, 3)
    uv = j33[..., :2] / j33[..., 2:] + noise * (1.0 - ckd)[..., None] * unitd
    j2dc = torch.cat([uv, ckd[..., None]], -1).contiguous()
    ori = torch.empty(B, T, 6, 3, 3, device=dev)
    acc = torch.empty(B, T, 6, 3, device=dev)
    for b in range(B):                                                                    # (the stencil runs along one sequence)
        o, a, _, _ = synthesize_imu(model, pose[b], tran[b], smooth_n=2 if T > 4 else 1)
        ori[b], acc[b] = o, a
    return {"j2dc": j2dc, "accc": acc, "oric": ori, "pose": pose, "tran": tran,
            "gravityc": np.stack(grav).astype(np.float32), "first_tran": np.stack(tr)[:, 0].astype(np.float32),
            "conf": np.stack(ck).mean(-1).astype(np.float32)}
