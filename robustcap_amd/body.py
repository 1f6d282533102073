"""Body-model constants of the path and the ``ParametricModel`` call surface on top of the HIP kernels.

Mirrors the part of ``articulate/model.py`` the sig_mp path uses (``ParametricModel``: L21-40 loading, L78-93
rest pose, L95-165 FK/IK wrappers, L209-241 ``forward_kinematics``), restricted to the 33 landmark vertices
``config.mp_mask`` that ``sync_mp3d`` gathers (net/sig_mp.py:287-299). All arithmetic runs in
``librobustcap_hip.so``; this file only prepares constants and marshals pointers.
"""
import ctypes as C
import pickle

import numpy as np
import torch

from . import _lib
from . import config as cfg


def load_smpl_pickle(path):
    """Read an official SMPL pickle into the dict layout used here (J, v_template, weights, parent).
    Same fields as articulate/model.py:29-39 (needs scipy for the pickled sparse J_regressor)."""
    with open(path, "rb") as f:
        data = pickle.load(f, encoding="latin1")
    parent = np.asarray(data["kintree_table"][0]).astype(np.int64).copy()
    parent[0] = -1
    body = {"J": np.asarray(data["J"], np.float32), "v_template": np.asarray(data["v_template"], np.float32),
            "weights": np.asarray(data["weights"], np.float32), "parent": parent}
    if "shapedirs" in data and "J_regressor" in data:                      # only needed for shape=... (model.py:33-35)
        jr = data["J_regressor"]
        body["J_regressor"] = np.asarray(jr.toarray() if hasattr(jr, "toarray") else jr, np.float32)
        body["shapedirs"] = np.asarray(data["shapedirs"], np.float32)[:, :, :10]
    return body


def body_arrays(body):
    """(parent int32[24], J f32[24,3], w33 f32[33,24], v33 f32[33,3]) for rc_set_body."""
    ids = list(cfg.mp_mask)
    parent = np.asarray(body["parent"]).astype(np.int32).copy()
    parent[0] = 0
    return (np.ascontiguousarray(parent), np.ascontiguousarray(body["J"], dtype=np.float32),
            np.ascontiguousarray(np.asarray(body["weights"], np.float32)[ids]),
            np.ascontiguousarray(np.asarray(body["v_template"], np.float32)[ids]))


def shaped_body(ctx, body, shape):
    """The body dict with the shape blendshapes applied: v = shapedirs . beta + v_template, J = J_regressor . v
    (articulate/model.py:88-91; the root alignment happens in rc_set_body like for the mean shape). ``shape``: 10 betas, or
    [n,10] with identical rows (a context holds ONE shape; ``ParametricModel.forward_kinematics`` groups frames of different
    shapes and calls this once per distinct row). Computed on the device (rc_shape_body)."""
    beta = torch.as_tensor(shape, dtype=torch.float32).detach().cpu().reshape(-1, 10)
    if not bool((beta == beta[:1]).all()):
        raise NotImplementedError("one shape per model / sequence: the rows of `shape` must be identical")
    if "shapedirs" not in body or "J_regressor" not in body:
        raise ValueError("this body has no shapedirs / J_regressor (needed for shape=...)")
    vt = np.ascontiguousarray(body["v_template"], dtype=np.float32)
    sd = np.ascontiguousarray(np.asarray(body["shapedirs"], np.float32)[:, :, :10])
    jr = np.ascontiguousarray(body["J_regressor"], dtype=np.float32)
    b = np.ascontiguousarray(beta[0].numpy())
    V = vt.shape[0]
    v, j = np.empty((V, 3), np.float32), np.empty((24, 3), np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    _lib.check(ctx, _lib.load().rc_shape_body(ctx, p(vt), p(sd), p(jr), p(b), V, p(v), p(j)), "rc_shape_body")
    out = dict(body)
    out["v_template"], out["J"] = v, j
    return out


def set_body(ctx, body):
    lib = _lib.load()
    parent, J, w33, v33 = body_arrays(body)
    rc = lib.rc_set_body(ctx, parent.ctypes.data_as(C.c_void_p), J.ctypes.data_as(C.c_void_p),
                         w33.ctypes.data_as(C.c_void_p), v33.ctypes.data_as(C.c_void_p))
    _lib.check(ctx, rc, "rc_set_body")


def _f32c(t, device):
    return t.to(device=device, dtype=torch.float32).contiguous()


class ParametricModel:
    """Drop-in for the methods of ``art.ParametricModel`` that the path calls; device tensors in and out."""

    def __init__(self, official_model_file=None, use_pose_blendshape=False, device="cuda", body=None):
        if use_pose_blendshape:
            raise NotImplementedError("pose blendshapes are off on the sig_mp path (model.py:237, default False)")
        self._body = body if body is not None else load_smpl_pickle(official_model_file)
        self._mean_body = self._body
        self._shape_key = None
        self.parent = [None] + [int(p) for p in self._body["parent"][1:]]
        self.device = torch.device(device)
        self._lib = _lib.load()
        self._ctx = C.c_void_p()
        _lib.check(None, self._lib.rc_create(1, 0, C.byref(self._ctx)), "rc_create")
        set_body(self._ctx, self._body)

    def __del__(self):
        if getattr(self, "_ctx", None):
            self._lib.rc_destroy(self._ctx)
            self._ctx = None

    def set_shape(self, shape=None):
        """Switch the model to the body of these shape parameters (None = mean shape): every method then works on it.
        The reference passes ``shape`` per call (model.py:209-229); here it is a state of the context (rc_set_body)."""
        key = None if shape is None else torch.as_tensor(shape, dtype=torch.float32).detach().cpu().reshape(-1, 10)[0].numpy().tobytes()
        if key == self._shape_key:
            return
        self._body = self._mean_body if shape is None else shaped_body(self._ctx, self._mean_body, shape)
        set_body(self._ctx, self._body)
        self._shape_key = key
        if self.__dict__.get("_mesh_set"):
            self._mesh_set = False
            self._ensure_mesh()

    def inverse_kinematics_R(self, R_global):
        Rg = _f32c(R_global, self.device).view(-1, 24, 3, 3)
        out = torch.empty_like(Rg)
        _lib.check(self._ctx, self._lib.rc_ik_r(self._ctx, _lib.ptr(Rg), _lib.ptr(out), Rg.shape[0], _lib.stream_ptr()), "rc_ik_r")
        return out

    def forward_kinematics_R(self, R_local):
        """articulate/model.py:131-145: global rotations from local ones (chained down the tree)."""
        Rl = _f32c(R_local, self.device).view(-1, 24, 3, 3)
        out = torch.empty_like(Rl)
        _lib.check(self._ctx, self._lib.rc_fk_r(self._ctx, _lib.ptr(Rl), _lib.ptr(out), Rl.shape[0], _lib.stream_ptr()), "rc_fk_r")
        return out

    def bone_vector_to_joint_position(self, bone_vec):
        """articulate/model.py:95-111: joint positions from parent->child bone vectors, [n,24,3]."""
        b = _f32c(bone_vec, self.device).view(-1, 24, 3)
        out = torch.empty_like(b)
        _lib.check(self._ctx, self._lib.rc_bone_to_joint(self._ctx, _lib.ptr(b), _lib.ptr(out), b.shape[0], _lib.stream_ptr()), "rc_bone_to_joint")
        return out

    def joint_position_to_bone_vector(self, joint_pos):
        """articulate/model.py:113-129: bone vectors (child - parent; root kept) from joint positions, [n,24,3]."""
        j = _f32c(joint_pos, self.device).view(-1, 24, 3)
        out = torch.empty_like(j)
        _lib.check(self._ctx, self._lib.rc_joint_to_bone(self._ctx, _lib.ptr(j), _lib.ptr(out), j.shape[0], _lib.stream_ptr()), "rc_joint_to_bone")
        return out

    def get_zero_pose_joint_and_vertex(self, shape=None):
        """articulate/model.py:78-93: (joints [24,3], vertices [V,3]) of the mean shape, root joint at the origin; with
        ``shape`` [batch,10] the reference's batched form ([batch,24,3], [batch,V,3]), one rc_shape_body per distinct row."""
        if shape is not None:
            beta = torch.as_tensor(shape, dtype=torch.float32).detach().cpu().reshape(-1, 10)
            if beta.shape[0] > 1 and not bool((beta == beta[:1]).all()):
                uniq, inv = torch.unique(beta, dim=0, return_inverse=True)
                parts = [self.get_zero_pose_joint_and_vertex(uniq[g]) for g in range(uniq.shape[0])]
                return (torch.stack([parts[int(g)][0] for g in inv]), torch.stack([parts[int(g)][1] for g in inv]))
            shape = beta[0]
        self.set_shape(shape)
        self._ensure_mesh()
        j = torch.empty(24, 3, device=self.device)
        v = torch.empty(self._V, 3, device=self.device)
        _lib.check(self._ctx, self._lib.rc_zero_pose(self._ctx, _lib.ptr(j), _lib.ptr(v), _lib.stream_ptr()), "rc_zero_pose")
        return j, v

    def bone_fk(self, R_global):
        """fk() of forward_online (net/sig_mp.py:131-135): joints from GLOBAL rotations, root at the origin."""
        Rg = _f32c(R_global, self.device).view(-1, 24, 3, 3)
        out = torch.empty(Rg.shape[0], 24, 3, device=self.device)
        _lib.check(self._ctx, self._lib.rc_fk_bone(self._ctx, _lib.ptr(Rg), _lib.ptr(out), Rg.shape[0], _lib.stream_ptr()), "rc_fk_bone")
        return out

    def forward_kinematics(self, pose, shape=None, tran=None, calc_mesh=False):
        """(global rotations, joints[, landmarks]). With ``calc_mesh`` the third output is the 33-landmark set
        ``sync_mp3d(vert, joint)`` -- the only part of the 6890-vertex mesh the path consumes.
        ``shape`` (model.py:209-229): None, 10 betas for every frame, or [batch, 10] with one row per frame -- the frames are
        then grouped by distinct beta row and every group runs on its own shaped body (``rc_shape_body`` once per distinct
        shape; a batch normally holds a handful of subjects)."""
        pose = _f32c(pose, self.device).view(-1, 24, 3, 3)
        n = pose.shape[0]
        if shape is not None:
            beta = torch.as_tensor(shape, dtype=torch.float32).detach().cpu().reshape(-1, 10)
            if beta.shape[0] not in (1, n):
                raise ValueError(f"shape must expand to [{n}, 10]")
            if beta.shape[0] == n and n > 1 and not bool((beta == beta[:1]).all()):
                uniq, inv = torch.unique(beta, dim=0, return_inverse=True)
                outs = None
                tr = None if tran is None else _f32c(tran, self.device).view(n, 3)
                for g in range(uniq.shape[0]):
                    idx = (inv == g).nonzero().flatten().to(self.device)
                    part = self.forward_kinematics(pose[idx], uniq[g], None if tr is None else tr[idx], calc_mesh)
                    if outs is None:
                        outs = [torch.empty((n,) + tuple(o.shape[1:]), device=self.device) for o in part]
                    for o, q in zip(outs, part):
                        o[idx] = q
                return tuple(outs)
            shape = beta[0]
        self.set_shape(shape)                                           # one shape for all of these frames (model.py:228)
        tran = torch.zeros(n, 3, device=self.device) if tran is None else _f32c(tran, self.device).view(n, 3)
        grot = torch.empty_like(pose)
        joint = torch.empty(n, 24, 3, device=self.device)
        j33 = torch.empty(n, 33, 3, device=self.device)
        _lib.check(self._ctx, self._lib.rc_body_fk(self._ctx, _lib.ptr(pose), _lib.ptr(tran), _lib.ptr(grot), _lib.ptr(joint),
                                                   _lib.ptr(j33), n, _lib.stream_ptr()), "rc_body_fk")
        return (grot, joint, j33) if calc_mesh else (grot, joint)

    def forward_mesh(self, pose, tran=None):
        """All V vertices of ``forward_kinematics(pose, tran=tran, calc_mesh=True)[2]`` (articulate/model.py:235-241),
        for the mesh metrics of evaluate.py:120-133. Returns a device tensor [n, V, 3]."""
        self._ensure_mesh()
        pose = _f32c(pose, self.device).view(-1, 24, 3, 3)
        n = pose.shape[0]
        tran = torch.zeros(n, 3, device=self.device) if tran is None else _f32c(tran, self.device).view(n, 3)
        vert = torch.empty(n, self._V, 3, device=self.device)
        _lib.check(self._ctx, self._lib.rc_body_mesh(self._ctx, _lib.ptr(pose), _lib.ptr(tran), _lib.ptr(vert), n, _lib.stream_ptr()),
                   "rc_body_mesh")
        return vert

    def _ensure_mesh(self):
        if not self.__dict__.get("_mesh_set"):
            vt = np.ascontiguousarray(self._body["v_template"], dtype=np.float32)
            w = np.ascontiguousarray(self._body["weights"], dtype=np.float32)
            _lib.check(self._ctx, self._lib.rc_set_mesh(self._ctx, vt.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p),
                                                        vt.shape[0]), "rc_set_mesh")
            self._mesh_set, self._V = True, vt.shape[0]

    def set_regressor(self, j_regressor, n_used=14):
        """Upload ``J_regressor_h36m`` [n_rows, V]; the metrics keep its first ``n_used`` rows (evaluate.py:122-125)."""
        self._ensure_mesh()
        Jr = np.ascontiguousarray(np.asarray(j_regressor, dtype=np.float32))
        if Jr.ndim != 2 or Jr.shape[1] != self._V:
            raise ValueError(f"j_regressor must be [n_rows, {self._V}]")
        _lib.check(self._ctx, self._lib.rc_set_regressor(self._ctx, Jr.ctypes.data_as(C.c_void_p), Jr.shape[0], int(n_used)), "rc_set_regressor")
        self._regressor_id = id(j_regressor)

    def mesh_metrics(self, pose, gt_pose):
        """evaluate.py:120-133 in one kernel: returns (per-frame [n,3] device tensor of MPJPE / PVE / PA-MPJPE,
        their three means as Python floats). Keypoints come from the regressor set by ``set_regressor`` (else the
        24 SMPL joints)."""
        self._ensure_mesh()
        pose = _f32c(pose, self.device).view(-1, 24, 3, 3)
        gt = _f32c(gt_pose, self.device).view(-1, 24, 3, 3)
        if gt.shape != pose.shape:
            raise ValueError("pose and gt_pose must have the same number of frames")
        n = pose.shape[0]
        per_frame = torch.empty(n, 3, device=self.device)
        mean = (C.c_double * 3)()
        _lib.check(self._ctx, self._lib.rc_mesh_metrics(self._ctx, _lib.ptr(pose), _lib.ptr(gt), n, _lib.ptr(per_frame), mean, _lib.stream_ptr()),
                   "rc_mesh_metrics")
        return per_frame, [float(v) for v in mean]

    def reprojection_residual(self, pose, tran, keypoints_2d, cam_k, sigma=100.0):
        """``TemporalSMPLify.get_fitting_loss`` (temporal_smplify.py:198-220): [T,33] robust reprojection loss."""
        pose = _f32c(pose, self.device).view(-1, 24, 3, 3)
        T = pose.shape[0]
        tran, kp, K = _f32c(tran, self.device).view(T, 3), _f32c(keypoints_2d, self.device).view(T, 33, 3), _f32c(cam_k, self.device).view(3, 3)
        loss = torch.empty(T, 33, device=self.device)
        _lib.check(self._ctx, self._lib.rc_reproj_residual(self._ctx, _lib.ptr(pose), _lib.ptr(tran), _lib.ptr(kp), _lib.ptr(K),
                                                           C.c_float(sigma), _lib.ptr(loss), T, _lib.stream_ptr()), "rc_reproj_residual")
        return loss


def r6d_to_rotation_matrix(r6d, device="cuda"):
    """art.math.r6d_to_rotation_matrix (articulate/math/angular.py:249-264) on the GPU."""
    x = _f32c(r6d, torch.device(device)).view(-1, 6)
    out = torch.empty(x.shape[0], 3, 3, device=x.device)
    lib = _lib.load()
    _lib.check(None, lib.rc_r6d_to_rotmat(_lib.ptr(x), _lib.ptr(out), x.shape[0], _lib.stream_ptr()), "rc_r6d_to_rotmat")
    return out


def axis_angle_to_rotation_matrix(a, device="cuda"):
    """art.math.axis_angle_to_rotation_matrix (articulate/math/angular.py:221-233) on the GPU: [n,3] -> [n,3,3]."""
    x = _f32c(a, torch.device(device)).view(-1, 3)
    out = torch.empty(x.shape[0], 3, 3, device=x.device)
    lib = _lib.load()
    _lib.check(None, lib.rc_axis_angle_to_rotmat(_lib.ptr(x), _lib.ptr(out), x.shape[0], _lib.stream_ptr()), "rc_axis_angle_to_rotmat")
    return out


def rotation_matrix_to_axis_angle(r, device="cuda"):
    """art.math.rotation_matrix_to_axis_angle (angular.py:236-246, cv2.Rodrigues loop in the reference): [n,3,3] -> [n,3]."""
    x = _f32c(r, torch.device(device)).view(-1, 3, 3)
    out = torch.empty(x.shape[0], 3, device=x.device)
    lib = _lib.load()
    _lib.check(None, lib.rc_rotmat_to_axis_angle(_lib.ptr(x), _lib.ptr(out), x.shape[0], _lib.stream_ptr()), "rc_rotmat_to_axis_angle")
    return out


def rotation_matrix_to_r6d(r, device="cuda"):
    """art.math.rotation_matrix_to_r6d (angular.py:267-274): [n,3,3] -> [n,6] (first two columns)."""
    x = _f32c(r, torch.device(device)).view(-1, 3, 3)
    out = torch.empty(x.shape[0], 6, device=x.device)
    _lib.check(None, _lib.load().rc_rotmat_to_r6d(_lib.ptr(x), _lib.ptr(out), x.shape[0], _lib.stream_ptr()), "rc_rotmat_to_r6d")
    return out


def angle_between(rot1, rot2, device="cuda"):
    """art.math.angle_between for rotation matrices (angular.py:128-141): [n] angles in radians."""
    a, b = _f32c(rot1, torch.device(device)).view(-1, 3, 3), _f32c(rot2, torch.device(device)).view(-1, 3, 3)
    if a.shape != b.shape:
        raise ValueError("rot1 and rot2 must hold the same number of rotations")
    out = torch.empty(a.shape[0], device=a.device)
    _lib.check(None, _lib.load().rc_angle_between(_lib.ptr(a), _lib.ptr(b), _lib.ptr(out), a.shape[0], _lib.stream_ptr()), "rc_angle_between")
    return out


def lerp(a, b, t, device="cuda"):
    """art.math.lerp (general.py:15-24) with a Python-float weight: a * (1 - t) + b * t, unclamped."""
    x, y = _f32c(a, torch.device(device)), _f32c(b, torch.device(device))
    if x.shape != y.shape:
        raise ValueError("a and b must have the same shape")
    out = torch.empty_like(x)
    _lib.check(None, _lib.load().rc_lerp(_lib.ptr(x), _lib.ptr(y), C.c_double(float(t)), _lib.ptr(out), x.numel(), _lib.stream_ptr()), "rc_lerp")
    return out


def normalize_tensor(x, dim=-1, return_norm=False, device="cuda"):
    """art.math.normalize_tensor (general.py:27-39) over the last dimension; the norm keeps that dimension (size 1)."""
    v = _f32c(x, torch.device(device))
    if dim not in (-1, v.dim() - 1):
        raise NotImplementedError("normalize_tensor: only the last dimension (the path's only use)")
    width = v.shape[-1]
    rows = v.numel() // max(width, 1)
    out = torch.empty_like(v)
    norm = torch.empty(v.shape[:-1] + (1,), device=v.device) if return_norm else None
    _lib.check(None, _lib.load().rc_normalize_rows(_lib.ptr(v), _lib.ptr(out), _lib.ptr(norm), rows, width, _lib.stream_ptr()), "rc_normalize_rows")
    return (out, norm) if return_norm else out


def normalize_keypoints(kp, device="cuda"):
    """The bbox normalisation forward_online applies to its keypoints (net/sig_mp.py:150-152, get_bbox_scale L277-284):
    [n,33,3] -> [n,33,3]."""
    x = _f32c(kp, torch.device(device)).view(-1, 33, 3)
    out = torch.empty_like(x)
    _lib.check(None, _lib.load().rc_bbox_normalise(_lib.ptr(x), _lib.ptr(out), x.shape[0], _lib.stream_ptr()), "rc_bbox_normalise")
    return out


def position_error(p, t, device="cuda"):
    """art.PositionErrorEvaluator()(p, t) (articulate/evaluator.py:100-129): mean Euclidean distance of n 3-D points."""
    a, b = _f32c(p, torch.device(device)).reshape(-1, 3), _f32c(t, torch.device(device)).reshape(-1, 3)
    if a.shape != b.shape:
        raise ValueError("p and t must hold the same number of points")
    d = torch.empty(a.shape[0], device=a.device)
    mean = C.c_double()
    lib = _lib.load()
    _lib.check(None, lib.rc_position_error(_lib.ptr(a), _lib.ptr(b), a.shape[0], _lib.ptr(d), C.byref(mean), _lib.stream_ptr()), "rc_position_error")
    return mean.value


def reconstruction_error(S1, S2, device="cuda"):
    """utils.reconstruction_error(S1, S2, reduction=None) (utils.py:189-203): per-frame mean point distance after
    Procrustes alignment of S1 [n,k,3] onto S2 [n,k,3]. Returns a device tensor [n]."""
    a, b = _f32c(S1, torch.device(device)), _f32c(S2, torch.device(device))
    if a.shape != b.shape or a.dim() != 3 or a.shape[2] != 3:
        raise ValueError("S1 and S2 must both be [n, k, 3]")
    err = torch.empty(a.shape[0], device=a.device)
    lib = _lib.load()
    _lib.check(None, lib.rc_procrustes_error(_lib.ptr(a), _lib.ptr(b), a.shape[0], a.shape[1], _lib.ptr(err), _lib.stream_ptr()), "rc_procrustes_error")
    return err
