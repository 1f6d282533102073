"""smplify on the GPU: the optimiser the reference's evaluate.py runs on every sequence (evaluate.py:89).

Mirrors ``net/smplify/run.py:smplify_runner`` and ``net/smplify/temporal_smplify.py:TemporalSMPLify``:
  * ``get_fitting_loss`` -> ``temporal_body_fitting_loss(output='reprojection')`` (temporal_smplify.py:198-220,
    losses.py:36-37,43-46): one wave-per-frame HIP kernel (rc_reproj_residual);
  * ``__call__`` (temporal_smplify.py:97-196): L-BFGS with a strong-Wolfe line search over the axis-angle pose and the
    translation of the WHOLE sequence. The closure (loss + analytic gradient) is two HIP kernels with one workgroup
    per frame (csrc/rc_smplify.hip); the optimiser logic is csrc/rc_lbfgs.h, a restatement of torch.optim.LBFGS;
  * the gate ``mean_k loss[0] > loss_threshold`` and the per-frame ``update`` mask (run.py:24-34).
Everything numerical is in librobustcap_hip.so; this file prepares constants and marshals pointers.
"""
import ctypes as C
import pickle

import numpy as np
import torch

from . import _lib
from . import body as _body
from . import config as cfg


def load_gmm_pickle(path):
    """gmm_08.pkl of the reference (net/smplify/prior.py:118-123): dict with means, covars, weights."""
    with open(path, "rb") as f:
        return pickle.load(f, encoding="latin1")


def prior_arrays(gmm):
    """(means[8,69], precisions[8,69,69], nll_weights[8]) as float32, the buffers MaxMixturePrior registers
    (net/smplify/prior.py:124-147): precisions = inv(covars), nll_weights = weights / (c * sqrt(det) / min sqrt(det))."""
    means = np.asarray(gmm["means"], dtype=np.float32)
    covs = np.asarray(gmm["covars"], dtype=np.float32)
    w = np.asarray(gmm["weights"])
    prec = np.stack([np.linalg.inv(c) for c in covs]).astype(np.float32)
    sqrdets = np.array([np.sqrt(np.linalg.det(c)) for c in np.asarray(gmm["covars"])])
    const = (2 * np.pi) ** (69 / 2.0)
    nllw = np.asarray(w / (const * (sqrdets / sqrdets.min())), dtype=np.float32)
    if means.shape != (8, 69) or prec.shape != (8, 69, 69) or nllw.shape != (8,):
        raise ValueError("the pose prior must have 8 components of dimension 69")
    return np.ascontiguousarray(means), np.ascontiguousarray(prec), np.ascontiguousarray(nllw)


class TemporalSMPLify:
    """Device-side optimiser state: body constants, pose prior, work buffers (reused across sequences)."""

    def __init__(self, body=None, smpl_file=None, gmm=None, gmm_file=None, device="cuda", use_head=False, shape=None):
        """``shape`` (10 betas, or [T,10] with identical rows; temporal_smplify.py:84-86): the body every closure evaluation,
        residual and landmark of this optimiser uses (``set_shape`` switches it later, None = mean shape)."""
        self.model = _body.ParametricModel(smpl_file, device=device, body=body)
        self._shaped, self._mean_model = False, None
        if shape is not None:
            self.model.set_shape(shape)
            self._shaped = True
        self.device = self.model.device
        self._lib, self._ctx = self.model._lib, self.model._ctx
        self.has_prior = False
        self.use_head = None
        self.set_use_head(use_head)
        if gmm is not None or gmm_file is not None:
            self.set_prior(gmm if gmm is not None else load_gmm_pickle(gmm_file))
        self.last_info = None

    def set_shape(self, shape=None):
        self.model.set_shape(shape)
        self._shaped = shape is not None

    def set_use_head(self, use_head):
        """temporal_smplify.py:92-94: the landmarks whose confidence is zeroed -- face + feet tips {1..9, 31, 32}, or only
        {31, 32} with ``use_head=True`` (the TotalCapture evaluation, evaluate.py:352)."""
        if use_head == self.use_head:
            return
        ids = (31, 32) if use_head else cfg.smplify_ignored_landmarks
        arr = (C.c_int32 * len(ids))(*ids)
        _lib.check(self._ctx, self._lib.rc_set_ignored_landmarks(self._ctx, arr, len(ids)), "rc_set_ignored_landmarks")
        self.use_head = bool(use_head)

    def set_prior(self, gmm):
        means, prec, nllw = prior_arrays(gmm)
        rc = self._lib.rc_smplify_set_prior(self._ctx, means.ctypes.data_as(C.c_void_p), prec.ctypes.data_as(C.c_void_p),
                                            nllw.ctypes.data_as(C.c_void_p))
        _lib.check(self._ctx, rc, "rc_smplify_set_prior")
        self.has_prior = True

    def get_fitting_loss(self, pose, tran, keypoints_2d, cam_k, sigma=100.0):
        return self.model.reprojection_residual(pose, tran, keypoints_2d, cam_k, sigma)

    def loss_and_grad(self, body_pose, tran, keypoints_2d, ref3d, imu_aa, cam_k):
        """One closure evaluation (temporal_smplify.py:150-166). body_pose [T,72] axis-angle, tran [T,3],
        keypoints_2d [T,33,3], ref3d [T,33,3], imu_aa [T,18]. Returns (loss float, grad_pose [T,72], grad_tran [T,3])."""
        dev = self.device
        T = body_pose.shape[0]
        x = torch.cat([_body._f32c(body_pose, dev).reshape(-1), _body._f32c(tran, dev).reshape(-1)]).contiguous()
        kp, ref, imu = (_body._f32c(a, dev) for a in (keypoints_2d, ref3d, imu_aa))
        K = np.ascontiguousarray(torch.as_tensor(cam_k).detach().cpu().numpy(), dtype=np.float32).reshape(9)
        grad = torch.empty_like(x)
        loss = C.c_double()
        rc = self._lib.rc_smplify_loss_grad(self._ctx, _lib.ptr(x), _lib.ptr(kp), _lib.ptr(ref), _lib.ptr(imu),
                                            K.ctypes.data_as(C.c_void_p), T, C.byref(loss), _lib.ptr(grad), _lib.stream_ptr())
        _lib.check(self._ctx, rc, "rc_smplify_loss_grad")
        return loss.value, grad[:T * 72].view(T, 72), grad[T * 72:].view(T, 3)

    def run(self, pose, tran, keypoints_2d, imu_ori, cam_k, lr=1.0, max_iter=20, loss_threshold=20000):
        """Pre-check + optimise + update mask on the device. Returns (pose [T,24,3,3], tran [T,3], update | None)
        as device tensors / a bool host tensor."""
        dev = self.device
        pose = _body._f32c(pose, dev).view(-1, 24, 3, 3)
        T = pose.shape[0]
        tran, kp, ori = _body._f32c(tran, dev).view(T, 3), _body._f32c(keypoints_2d, dev).view(T, 33, 3), _body._f32c(imu_ori, dev).view(T, 6, 3, 3)
        K = np.ascontiguousarray(torch.as_tensor(cam_k).detach().cpu().numpy(), dtype=np.float32).reshape(9)
        pose_out, tran_out = torch.empty_like(pose), torch.empty_like(tran)
        update = np.zeros(T, dtype=np.uint8)
        info = _lib.RcSmplifyInfo()
        ref3d = None
        if self._shaped:     # reference quirk: the preserved 3D landmarks come from the MEAN-shape body (temporal_smplify.py:112)
            if self._mean_model is None:
                self._mean_model = _body.ParametricModel(device=dev, body=self.model._mean_body)
            ref3d = self._mean_model.forward_kinematics(pose, tran=tran, calc_mesh=True)[2].contiguous()
            torch.cuda.current_stream().synchronize()
            _lib.check(self._ctx, self._lib.rc_smplify_set_ref3d(self._ctx, _lib.ptr(ref3d)), "rc_smplify_set_ref3d")
        rc = self._lib.rc_smplify_run(self._ctx, _lib.ptr(pose), _lib.ptr(tran), _lib.ptr(kp), _lib.ptr(ori), K.ctypes.data_as(C.c_void_p),
                                      T, C.c_float(lr), int(max_iter), C.c_float(loss_threshold), _lib.ptr(pose_out), _lib.ptr(tran_out),
                                      update.ctypes.data_as(C.c_void_p), C.byref(info), _lib.stream_ptr())
        _lib.check(self._ctx, rc, "rc_smplify_run")
        self.last_info = {k: getattr(info, k) for k, _ in info._fields_ if k != "reserved"}
        return pose_out, tran_out, (torch.from_numpy(update.astype(bool)) if info.status == 1 else None)


    def run_batch(self, rows, lr=1.0, max_iter=20, loss_threshold=20000):
        """The same as ``run`` for a list of independent rows ``(pose, tran, keypoints_2d, imu_ori, cam_k)`` at once
        (evaluate.py:86-90 loops them): the rows' optimisers advance in lock-step rounds on the device, one launch per kind of
        operation over all rows (rc_smplify_run_batch). Returns a list of (pose, tran, update | None) and fills
        ``last_batch_info`` (one dict per row). Mean shape only (``set_shape`` rows go through ``run``)."""
        if self._shaped:
            raise NotImplementedError("run_batch: rows with shape=... go through run() one at a time")
        dev = self.device
        n = len(rows)
        prep, Ks = [], np.empty((n, 9), np.float32)
        for r, (pose, tran, kp, ori, cam_k) in enumerate(rows):
            pose = _body._f32c(pose, dev).view(-1, 24, 3, 3)
            T = pose.shape[0]
            prep.append((pose, _body._f32c(tran, dev).view(T, 3), _body._f32c(kp, dev).view(T, 33, 3), _body._f32c(ori, dev).view(T, 6, 3, 3),
                         torch.empty_like(pose), torch.empty(T, 3, device=dev), np.zeros(T, dtype=np.uint8)))
            Ks[r] = np.asarray(torch.as_tensor(cam_k).detach().cpu().numpy(), np.float32).reshape(9)
        Tn = (C.c_int64 * n)(*[p[0].shape[0] for p in prep])
        arr = lambda k: (C.c_void_p * n)(*[p[k].data_ptr() for p in prep])
        upd = (C.c_void_p * n)(*[p[6].ctypes.data for p in prep])
        infos = (_lib.RcSmplifyInfo * n)()
        rc = self._lib.rc_smplify_run_batch(self._ctx, n, Tn, arr(0), arr(1), arr(2), arr(3), Ks.ctypes.data_as(C.c_void_p), C.c_float(lr),
                                            int(max_iter), C.c_float(loss_threshold), arr(4), arr(5), upd, infos, _lib.stream_ptr())
        _lib.check(self._ctx, rc, "rc_smplify_run_batch")
        self.last_batch_info = [{("rounds" if k == "reserved" else k): getattr(infos[r], k) for k, _ in infos[r]._fields_} for r in range(n)]
        return [(p[4], p[5], torch.from_numpy(p[6].astype(bool)) if infos[r].status == 1 else None) for r, p in enumerate(prep)]


ResidualRunner = TemporalSMPLify        # earlier name of the residual-only object


def smplify_runner(pred_pose, pred_tran, j2dc, imu_ori, batch_size, cam_k, lr=1.0, opt_steps=1, use_lbfgs=True,
                   loss_threshold=20000, shape=None, use_head=False, runner=None, body=None, gmm=None):
    """Same signature and return convention as net/smplify/run.py:smplify_runner.

    pred_pose [T,24,3,3], pred_tran [T,3], j2dc [T,33,3] in pixels, imu_ori [T,6,3,3] in the camera frame,
    cam_k [3,3]. Returns (pose [T,24,3,3] cpu, tran [T,3] cpu, update) with update None if the sequence failed the
    pre-check. ``runner`` (a TemporalSMPLify with the prior set) is reused across calls; otherwise ``body`` and
    ``gmm`` build one."""
    if not use_lbfgs:
        # The reference's own Adam branch cannot run: temporal_smplify.py:171-175 hands the [T,72] axis-angle parameters to
        # forward_kinematics as if they were rotation matrices (view(T,-1,3,3) -> 8 "joints") and torch.cat raises
        # "Sizes of tensors must match" (verified on the imported reference). Nothing to be equal to.
        raise NotImplementedError("use_lbfgs=False raises inside the reference itself (temporal_smplify.py:171-175); only L-BFGS is built")
    if opt_steps != 1:
        raise NotImplementedError("the reference only runs opt_steps=1 (evaluate.py:89)")
    runner = runner or TemporalSMPLify(body=body, gmm=gmm)
    runner.set_use_head(use_head)
    runner.set_shape(shape)                                       # temporal_smplify.py:84-86,158-159: None = mean shape
    if not runner.has_prior:
        raise _lib.RobustcapLibraryError("smplify_runner needs the GMM pose prior (gmm= or TemporalSMPLify.set_prior)")
    T = int(batch_size)
    pose, tran, update = runner.run(pred_pose.reshape(T, 24, 3, 3), pred_tran.reshape(T, 3), j2dc.reshape(T, 33, 3),
                                    imu_ori.reshape(T, 6, 3, 3), cam_k, lr=lr, loss_threshold=loss_threshold)
    return pose.cpu().reshape(-1, 24, 3, 3), tran.cpu().reshape(-1, 3), update
