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
