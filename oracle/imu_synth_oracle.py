"""CPU oracle for the IMU synthesis -- TEST INFRASTRUCTURE, NOT PRODUCT.

numpy/torch-CPU restatement of preprocess.py:22-33 (``_syn_acc``) and of the recipe at preprocess.py:206-214
(FK with mesh -> ``imu_ori = gp[:, ji_mask]``, ``imu_acc = _syn_acc(vert[:, vi_mask])``).
Pinned by tests/test_imu_synth_oracle.py against tests/golden/imu_synth.npz (captured from the reference itself).
"""
import numpy as np
import torch

from robustcap_amd import config as C
from . import sig_mp_oracle as O


def syn_acc(v, smooth_n=2):
    """v [T, ...] positions at 60 fps -> accelerations [T, ...]: second differences * 3600, zero at both ends; for
    smooth_n > 1 the interior [n:-n] is overwritten by the wide stencil (v[i] + v[i+2n] - 2 v[i+n]) * 3600 / n^2."""
    v = np.asarray(v, np.float32)
    T = v.shape[0]
    if T < 2 * smooth_n + 1 and smooth_n // 2 != 0:
        raise ValueError("the reference needs at least 2 * smooth_n + 1 frames")
    acc = np.zeros_like(v)
    acc[1:-1] = (v[:-2] + v[2:] - np.float32(2) * v[1:-1]) * np.float32(3600)
    if smooth_n // 2 != 0:
        n = smooth_n
        acc[n:-n] = (v[:-2 * n] + v[2 * n:] - np.float32(2) * v[n:-n]) * np.float32(3600) / np.float32(n ** 2)
    return acc


def synthesize_imu(body, pose_aa, tran, smooth_n=2):
    """(imu_ori [T,6,3,3], imu_acc [T,6,3], joint3d [T,24,3], vert6 [T,6,3]) in the frame of pose/tran."""
    ob = O.OracleBody(body, vertex_ids=C.vi_mask)
    T = pose_aa.shape[0]
    p = O.axis_angle_to_rotation_matrix(torch.as_tensor(pose_aa, dtype=torch.float32).reshape(-1, 3)).view(T, 24, 3, 3)
    gp, joint, vert = ob.forward_kinematics(p, torch.as_tensor(tran, dtype=torch.float32))
    return gp[:, list(C.ji_mask)].numpy(), syn_acc(vert.numpy(), smooth_n), joint.numpy(), vert.numpy()
