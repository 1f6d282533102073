"""CPU oracle for the evaluation metrics -- TEST INFRASTRUCTURE, NOT PRODUCT.

numpy/torch-CPU restatement of
  evaluate.py:120-133                  cal_mpjpe: full-mesh FK, J_regressor, pelvis alignment, MPJPE / PVE / PA-MPJPE
  utils.py:138-203                     compute_similarity_transform(_batch), reconstruction_error (Procrustes via SVD)
  articulate/evaluator.py:100-129      PositionErrorEvaluator
Pinned by tests/test_metrics_oracle.py against tests/golden/metrics.npz (captured from the reference itself by
oracle/capture_metrics.py with a synthetic body and a synthetic J_regressor).
"""
import numpy as np
import torch

from . import sig_mp_oracle as O


def similarity_transform(S1, S2):
    """S1, S2 [N,3] -> S1 aligned onto S2 by the optimal scale * rotation + translation (utils.py:138-187)."""
    S1, S2 = np.asarray(S1).T, np.asarray(S2).T
    mu1, mu2 = S1.mean(axis=1, keepdims=True), S2.mean(axis=1, keepdims=True)
    X1, X2 = S1 - mu1, S2 - mu2
    var1 = np.sum(X1 ** 2)
    K = X1.dot(X2.T)
    U, s, Vh = np.linalg.svd(K)
    V = Vh.T
    Z = np.eye(3)
    Z[-1, -1] *= np.sign(np.linalg.det(U.dot(V.T)))
    R = V.dot(Z.dot(U.T))
    scale = np.trace(R.dot(K)) / var1
    t = mu2 - scale * (R.dot(mu1))
    return (scale * R.dot(S1) + t).T


def reconstruction_error(S1, S2):
    """per-frame mean joint distance after Procrustes alignment (utils.py:189-203, reduction=None)."""
    hat = np.stack([similarity_transform(a, b) for a, b in zip(S1, S2)])
    return np.sqrt(((hat - S2) ** 2).sum(axis=-1)).mean(axis=-1)


def frame_metrics(body, j_regressor, pose, gt_pose):
    """per-frame (mpjpe, pve, pa_mpjpe), each [T]: the three means of cal_mpjpe before the final .mean()."""
    ob = O.OracleBody(body, vertex_ids=range(body["v_template"].shape[0]))
    zero = torch.zeros(pose.shape[0], 3)
    vt = ob.forward_kinematics(gt_pose, zero)[2]
    vp = ob.forward_kinematics(pose, zero)[2]
    Jr = torch.as_tensor(j_regressor, dtype=torch.float32)
    kt, kp = torch.matmul(Jr, vt)[:, :14], torch.matmul(Jr, vp)[:, :14]
    kt, kp = kt - kt[:, :1], kp - kp[:, :1]
    return ((kt - kp).norm(dim=2).mean(dim=1).numpy(), (vt - vp).norm(dim=2).mean(dim=1).numpy(),
            reconstruction_error(kp.numpy(), kt.numpy()))


def cal_mpjpe(body, j_regressor, pose, gt_pose):
    return np.array([m.mean() for m in frame_metrics(body, j_regressor, pose, gt_pose)], np.float64)


def position_error(p, t):
    d = np.asarray(p, np.float32).reshape(-1, 3) - np.asarray(t, np.float32).reshape(-1, 3)
    return float(np.sqrt((d.astype(np.float32) ** 2).sum(axis=1)).mean())
