"""TEST INFRASTRUCTURE ONLY (see oracle/sig_mp_oracle.py): torch-CPU restatement of the reference's evaluation-loop input
preparation, evaluate.py:38-51,70-73 (evaluate_aist_ours). Imported by tests only; the product path (robustcap_amd.evaluate)
runs rc_camera_inputs_rows on the device and never touches this module. Parity: pinned by
tests/test_gpu_evaluate.py::test_camera_inputs_match_reference_formulas (same formulas, HIP kernel vs this arithmetic)."""
import torch


def camera_inputs(kp_norm, imu_acc_w, imu_ori_w, K, Tcw, image_size=(1920, 1080)):
    """One (sequence, camera): j2dc [T,33,3], accc [T,6,3], oric [T,6,3,3], gravityc [3] in the camera frame."""
    kp = torch.as_tensor(kp_norm, dtype=torch.float32).clone()
    kp[..., 0] *= image_size[0]                                              # evaluate.py:43-44
    kp[..., 1] *= image_size[1]
    K = torch.as_tensor(K, dtype=torch.float32).reshape(3, 3)
    Tcw = torch.as_tensor(Tcw, dtype=torch.float32).reshape(4, 4)
    R = Tcw[:3, :3]
    ones = torch.cat((kp[..., :2], torch.ones_like(kp[..., :1])), -1)
    j2dc = (K.inverse() @ ones.unsqueeze(-1)).squeeze(-1)                    # evaluate.py:70-71
    j2dc[..., -1] = kp[..., -1]                                              # evaluate.py:72
    oric = R @ torch.as_tensor(imu_ori_w, dtype=torch.float32)               # evaluate.py:38
    accc = torch.as_tensor(imu_acc_w, dtype=torch.float32) @ R.T             # evaluate.py:39
    g = R @ torch.tensor([0.0, -1.0, 0.0])                                   # evaluate.py:73
    return j2dc, accc, oric, g


def first_translation(tran_w, Tcw):
    """label translation of frame 0 in the camera frame, evaluate.py:46-49,77."""
    Tcw = torch.as_tensor(Tcw, dtype=torch.float32).reshape(4, 4)
    return (torch.as_tensor(tran_w, dtype=torch.float32) @ Tcw[:3, :3].T + Tcw[:3, 3])[0]
