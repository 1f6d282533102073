"""smplify forward residual on the GPU: the pre-check of ``smplify_runner`` (net/smplify/run.py:6-34).

What is built: ``TemporalSMPLify.get_fitting_loss`` ->
``temporal_body_fitting_loss(output='reprojection')`` (temporal_smplify.py:198-220, losses.py:36-37,43-46) as
one wave-per-frame HIP kernel (rc_reproj_residual), and the gate ``mean_k loss[0] > loss_threshold`` (run.py:27-29).

What is NOT built yet (section 8(f) rank 1): the L-BFGS optimiser behind the gate (temporal_smplify.py:97-196).
``smplify_runner`` therefore returns the network prediction unchanged with ``update`` all-False when the gate
lets the sequence through, and -- exactly like the reference -- ``update=None`` when the gate rejects it.
"""
import torch

from . import body as _body


class ResidualRunner:
    """Holds the body constants on the device; ``get_fitting_loss`` mirrors the reference method."""

    def __init__(self, body=None, smpl_file=None, device="cuda"):
        self.model = _body.ParametricModel(smpl_file, device=device, body=body)

    def get_fitting_loss(self, pose, tran, keypoints_2d, cam_k, sigma=100.0):
        return self.model.reprojection_residual(pose, tran, keypoints_2d, cam_k, sigma)


def smplify_runner(pred_pose, pred_tran, j2dc, imu_ori, batch_size, cam_k, lr=1.0, opt_steps=1, use_lbfgs=True,
                   loss_threshold=20000, shape=None, use_head=False, runner=None, body=None):
    """Same signature and return convention as net/smplify/run.py:smplify_runner.

    pred_pose [T,24,3,3], pred_tran [T,3], j2dc [T,33,3] in pixels, cam_k [3,3]. Returns
    (pose [T,24,3,3] cpu, tran [T,3] cpu, update) with update None if the sequence failed the pre-check."""
    if shape is not None or use_head:
        raise NotImplementedError("shape / use_head variants are outside the built path (mean shape, ignored head landmarks)")
    runner = runner or ResidualRunner(body=body)
    T = int(batch_size)
    pose = pred_pose.reshape(T, 24, 3, 3)
    tran = pred_tran.reshape(T, 3)
    loss = runner.get_fitting_loss(pose, tran, j2dc.reshape(T, 33, 3), cam_k)       # [T,33] on the device
    opt_joint_loss = loss.mean(dim=-1)
    if float(opt_joint_loss[0].cpu()) > loss_threshold:                               # run.py:27-29
        return pose.cpu().reshape(-1, 24, 3, 3), tran.cpu().reshape(-1, 3), None
    update = torch.zeros(T, dtype=torch.bool)                                         # optimiser not built: nothing improves
    return pose.cpu().reshape(-1, 24, 3, 3), tran.cpu().reshape(-1, 3), update
