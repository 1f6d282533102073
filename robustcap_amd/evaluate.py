"""Evaluation harness for the sig_mp path: the counterpart of the reference's ``evaluate_aist_ours`` loop
(evaluate.py:20-117) with every (sequence, camera) a row of one batched, GPU-sharded run.

  dataset layout  : the reference's preprocessed ``test.pt`` dict (preprocess.py:229-237) -- per sequence ``pose``
                    [T,72], ``tran`` [T,3], ``imu_ori`` [T,6,3,3], ``imu_acc`` [T,6,3] in the world frame, per camera
                    ``cam_K``, ``cam_T`` (T_cw), ``joint2d_mp`` [n_cam,T,33,3] normalised by the image size.
                    ``robustcap_amd.synth.make_dataset`` generates the same layout (the real data is external).
  input prep      : rc_camera_inputs kernel = evaluate.py:38-51,70-73 (K^-1 [u,v,1], R_cw IMU, gravity).
  per-frame loop  : ``Net.forward_sequence`` -- all rows, all frames, one call (evaluate.py:75-83 is a Python loop).
  sharding        : contiguous (sequence, camera) blocks per rank, one final all-gather (robustcap_amd.dist).
  metrics         : root-position error (articulate/evaluator.py:100-129 semantics), root-aligned MPJPE over the 24
                    SMPL joints and global joint-angle error. The reference's H36M-14 MPJPE / PVE / PA-MPJPE need
                    the full 6890-vertex mesh and ``J_regressor_h36m.npy`` (absent): SURVEY.md section 8(f) rank 2.
"""
import os
import ctypes as C

import numpy as np
import torch

from . import _lib
from . import body as _body
from . import dist as rdist
from .net.sig_mp import Net


def camera_inputs(kp_norm, imu_acc_w, imu_ori_w, K, Tcw, image_size=(1920, 1080), device="cuda"):
    """evaluate.py:38-51 + 70-73 for one (sequence, camera). Returns device tensors j2dc [T,33,3], accc [T,6,3],
    oric [T,6,3,3] and the host gravity vector [3]."""
    dev = torch.device(device)
    kp = torch.as_tensor(kp_norm, dtype=torch.float32).clone()
    kp[..., 0] *= image_size[0]                                    # evaluate.py:43-44
    kp[..., 1] *= image_size[1]
    kp = kp.to(dev).contiguous()
    T = kp.shape[0]
    acc = torch.as_tensor(imu_acc_w, dtype=torch.float32).to(dev).contiguous()
    ori = torch.as_tensor(imu_ori_w, dtype=torch.float32).to(dev).contiguous()
    K = np.ascontiguousarray(np.asarray(K, np.float32).reshape(3, 3))
    Tcw = np.ascontiguousarray(np.asarray(Tcw, np.float32).reshape(4, 4))
    j2dc = torch.empty(T, 33, 3, device=dev)
    accc = torch.empty(T, 6, 3, device=dev)
    oric = torch.empty(T, 6, 3, 3, device=dev)
    g = np.zeros(3, np.float32)
    lib = _lib.load()
    rc = lib.rc_camera_inputs(_lib.ptr(kp), _lib.ptr(acc), _lib.ptr(ori), K.ctypes.data_as(C.c_void_p), Tcw.ctypes.data_as(C.c_void_p),
                              _lib.ptr(j2dc), _lib.ptr(accc), _lib.ptr(oric), g.ctypes.data_as(C.c_void_p), T, _lib.stream_ptr())
    _lib.check(None, rc, "rc_camera_inputs")
    torch.cuda.current_stream().synchronize()
    return j2dc, accc, oric, torch.from_numpy(g)


def rows_of(dataset):
    """[(sequence index, camera index)] in the reference's order (sequence-major, evaluate.py:32-33)."""
    return [(i, j) for i in range(len(dataset["pose"])) for j in range(len(dataset["cam_K"][i]))]


def labels(dataset, i, j, device="cuda"):
    """camera-frame ground truth of row (i, j): pose [T,24,3,3] with the root rotated by R_cw, tran = T_cw [tran;1]
    (evaluate.py:46-49)."""
    Tcw = torch.as_tensor(dataset["cam_T"][i][j], dtype=torch.float32)
    pose = _body.axis_angle_to_rotation_matrix(torch.as_tensor(dataset["pose"][i]).reshape(-1, 3), device).view(-1, 24, 3, 3).cpu()
    pose[:, 0] = Tcw[:3, :3] @ pose[:, 0]
    tran = torch.as_tensor(dataset["tran"][i], dtype=torch.float32) @ Tcw[:3, :3].T + Tcw[:3, 3]
    return pose, tran


def camera_inputs_rows(dataset, rows, Tmax, image_size=(1920, 1080), device="cuda"):
    """evaluate.py:38-51 + 70-73 for ALL the given (sequence, camera) rows in ONE kernel launch (rc_camera_inputs_rows):
    the dataset arrays are stacked on the host (padded to Tmax), uploaded once, and every row's frames are prepared on the
    device -- no per-row launches, copies or synchronisation. Returns device tensors j2dc [n,Tmax,33,3], accc [n,Tmax,6,3],
    oric [n,Tmax,6,3,3] (padding frames: zero keypoints / accelerations, identity orientations) and host gravity [n,3]."""
    dev = torch.device(device)
    n = len(rows)
    seqs = sorted({i for i, _ in rows})
    slot = {i: k for k, i in enumerate(seqs)}
    kp = np.zeros((n, Tmax, 33, 3), np.float32)
    acc = np.zeros((len(seqs), Tmax, 6, 3), np.float32)
    ori = np.zeros((len(seqs), Tmax, 6, 3, 3), np.float32)
    for i in seqs:
        T = len(dataset["pose"][i])
        acc[slot[i], :T] = np.asarray(dataset["imu_acc"][i], np.float32)
        ori[slot[i], :T] = np.asarray(dataset["imu_ori"][i], np.float32)
    K = np.zeros((n, 3, 3), np.float32)
    Tcw = np.zeros((n, 4, 4), np.float32)
    lens = np.zeros(n, np.int32)
    for r, (i, j) in enumerate(rows):
        T = len(dataset["pose"][i])
        kp[r, :T] = np.asarray(dataset["joint2d_mp"][i][j], np.float32)
        K[r], Tcw[r], lens[r] = np.asarray(dataset["cam_K"][i][j], np.float32), np.asarray(dataset["cam_T"][i][j], np.float32), T
    seq_of_row = np.asarray([slot[i] for i, _ in rows], np.int32)
    up = lambda a: torch.from_numpy(a).to(dev)
    kp_d, acc_d, ori_d, sor_d, len_d = up(kp), up(acc), up(ori), up(seq_of_row), up(lens)
    j2dc = torch.empty(n, Tmax, 33, 3, device=dev)
    accc = torch.empty(n, Tmax, 6, 3, device=dev)
    oric = torch.empty(n, Tmax, 6, 3, 3, device=dev)
    grav = np.zeros((n, 3), np.float32)
    scratch = torch.empty(n * 18, device=dev)                                   # 72 bytes of camera constants per row
    lib = _lib.load()
    rc = lib.rc_camera_inputs_rows(_lib.ptr(kp_d), _lib.ptr(acc_d), _lib.ptr(ori_d), _lib.ptr(sor_d), _lib.ptr(len_d),
                                   K.ctypes.data_as(C.c_void_p), Tcw.ctypes.data_as(C.c_void_p), float(image_size[0]), float(image_size[1]),
                                   n, Tmax, _lib.ptr(j2dc), _lib.ptr(accc), _lib.ptr(oric), grav.ctypes.data_as(C.c_void_p),
                                   _lib.ptr(scratch), _lib.stream_ptr())
    _lib.check(None, rc, "rc_camera_inputs_rows")
    torch.cuda.current_stream().synchronize()                                   # inputs of the launch stay alive until it ran
    return j2dc, accc, oric, torch.from_numpy(grav)


def first_translations(dataset, rows):
    """label translation of frame 0 of every row in its camera frame: T_cw [tran; 1] (evaluate.py:46-49,77) -- the same
    float32 expression as ``labels`` (so a row of the batched run equals that row run alone, bit for bit), without the pose."""
    ft = torch.zeros(len(rows), 3)
    for r, (i, j) in enumerate(rows):
        Tcw = torch.as_tensor(dataset["cam_T"][i][j], dtype=torch.float32)
        ft[r] = (torch.as_tensor(dataset["tran"][i], dtype=torch.float32) @ Tcw[:3, :3].T + Tcw[:3, 3])[0]
    return ft


def run_dataset(dataset, state_dict, body, use_first_tran=True, use_flat_floor=True, rows=None, device="cuda",
                run_smplify=False, gmm=None, smplify_info=None, image_size=(1920, 1080), smplify_workers=4, nets=None,
                gemm_mode=None):
    """Run every (sequence, camera) row of ``dataset`` (or the given subset) through the net; rows are sharded over
    the ranks of the initialised process group and gathered. Returns {(i, j): (pose [T,24,3,3], tran [T,3])} on the CPU.

    Input preparation is one kernel launch for all rows (``camera_inputs_rows``), the net one ``forward_sequence`` call.
    run_smplify=True refines every row with the smplify optimiser exactly where evaluate.py:86-90 does: after the
    net, on the pixel keypoints and the camera-frame IMU orientations of that row, lr=0.001, and -- like the reference
    -- takes the optimised pose and translation whether or not ``update`` says they improved. Rows are independent
    optimisation problems (the reference runs them one after another): ``smplify_workers`` host threads each drive their
    own optimiser context and HIP stream, so the line searches of several rows overlap on the device. ``gmm`` is the pose
    prior (dict means/covars/weights); ``smplify_info`` (a dict) receives the per-row optimiser records. ``nets``: an optional
    dict the caller keeps between calls -- the ``Net`` of a given row count is then built (weights re-packed and uploaded,
    ~0.4 s) once and only reset for the next dataset (an entry is reused only for the very same ``state_dict`` object).
    ``gemm_mode``: product arithmetic of the GEMMs (False fp32 MFMA, True split-bf16 products); None picks the library's
    default for the TOTAL number of rows of the run, not for this rank's shard, so a row's result does not depend on
    the number of ranks (``Net.default_gemm_mode``)."""
    all_rows = rows_of(dataset) if rows is None else list(rows)
    Tmax = max(len(dataset["pose"][i]) for i, _ in all_rows)
    split = Net.default_gemm_mode(len(all_rows)) if gemm_mode is None else bool(gemm_mode)

    def compute(mine):
        n = len(mine)
        j2d, acc, ori, grav = camera_inputs_rows(dataset, mine, Tmax, image_size=image_size, device=device)
        ft = first_translations(dataset, mine)
        hit = None if nets is None else nets.get(n)
        net = hit[0] if hit is not None and hit[1] is state_dict else None      # identity, not id(): ids are reused after gc
        if net is None:
            net = Net(body=body, batch=n, device=device)
            net.load_state_dict(state_dict)
            if nets is not None:
                nets[n] = (net, state_dict)
        else:
            net.reset_states()
        if bool(net.gemm_mode) != bool(split):                  # (only when it differs: a switch drops a captured live frame)
            net.set_gemm_mode(split)
        net.use_flat_floor = use_flat_floor
        net.gravityc = grav
        out_p, out_t = net.forward_sequence(j2d, acc, ori, first_tran=ft if use_first_tran else None, first_frame=not use_first_tran)
        if run_smplify:
            _refine_rows(dataset, mine, body, gmm, out_p, out_t, ori, image_size, device, smplify_info, smplify_workers)
        return out_p, out_t

    rows_out, out_p, out_t = shard_rows(all_rows, Tmax, compute, device=device)
    res = {}
    for r, (i, j) in enumerate(rows_out):
        T = len(dataset["pose"][i])
        res[(i, j)] = (out_p[r, :T], out_t[r, :T])
    return res


def shard_rows(all_rows, Tmax, compute, device="cuda"):
    """The multi-GPU shape of an evaluation (BASELINE config 3): contiguous blocks of the (sequence, camera) rows per rank
    (``dist.shard_range``), ``compute(rows of this rank) -> (pose [n,Tmax,24,3,3], tran [n,Tmax,3])`` on this rank's
    device, and ONE gather of the outputs -- the path's only collective (RCCL over xGMI; gloo on CPU hosts). A rank whose
    block is empty (fewer rows than ranks) computes nothing and still joins the gather. Returns (rows in output order,
    pose, tran) on the CPU: all rows on every rank when distributed, else this process's rows."""
    rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
    world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
    a, b = rdist.shard_range(len(all_rows), rank, world)
    mine = all_rows[a:b]
    n = len(mine)
    if n:
        out_p, out_t = compute(mine)
    else:
        out_p, out_t = torch.zeros(0, Tmax, 24, 3, 3, device=device), torch.zeros(0, Tmax, 3, device=device)
    if world > 1:
        # explicit widths: a rank whose shard is empty (rows < world) must still reach the collective
        cap_p = rdist.gather_rows(out_p[:n].reshape(n, Tmax * 216), len(all_rows))
        cap_t = rdist.gather_rows(out_t[:n].reshape(n, Tmax * 3), len(all_rows))
        out_p, out_t, rows_out = cap_p.view(-1, Tmax, 24, 3, 3), cap_t.view(-1, Tmax, 3), all_rows
    else:
        rows_out = mine
    return rows_out, out_p.cpu(), out_t.cpu()                           # one D2H for everything


def _refine_rows(dataset, mine, body, gmm, out_p, out_t, ori, image_size, device, smplify_info, workers):
    """smplify over the rows of this rank (evaluate.py:86-90 loops them one after another): ONE batched call -- every row's
    optimiser advances in lock-step rounds on the device (TemporalSMPLify.run_batch / rc_smplify_run_batch). ``workers``
    (run_dataset's ``smplify_workers``) is only read by round 3's scheme -- that many host threads, a context and a stream each --
    which the environment switch RC_SMPLIFY_WORKERS=n (n > 0, A/B runs) selects; the batched call has no worker count."""
    from .smplify import TemporalSMPLify
    torch.cuda.synchronize()
    scale = torch.tensor([float(image_size[0]), float(image_size[1]), 1.0])
    threads = int(os.environ.get("RC_SMPLIFY_WORKERS", "0"))                  # > 0: the thread-per-row scheme (A/B runs)
    if threads <= 0:
        runner = TemporalSMPLify(body=body, gmm=gmm, device=device)
        if not runner.has_prior:
            raise ValueError("run_smplify=True needs the GMM pose prior (gmm=)")
        rows = []
        for r, (i, j) in enumerate(mine):
            T = len(dataset["pose"][i])
            kp_pix = torch.as_tensor(dataset["joint2d_mp"][i][j], dtype=torch.float32) * scale      # evaluate.py:43-44 -> :87
            rows.append((out_p[r, :T], out_t[r, :T], kp_pix, ori[r, :T], dataset["cam_K"][i][j]))
        res = runner.run_batch(rows, lr=0.001)
        for r, (i, j) in enumerate(mine):
            T = len(dataset["pose"][i])
            out_p[r, :T], out_t[r, :T] = res[r][0], res[r][1]
            if smplify_info is not None:
                smplify_info[(i, j)] = dict(runner.last_batch_info[r])
        torch.cuda.synchronize()
        return
    import threading
    workers = max(1, min(threads if threads > 0 else int(workers or 1), len(mine)))
    runners = [TemporalSMPLify(body=body, gmm=gmm, device=device) for _ in range(workers)]
    if not runners[0].has_prior:
        raise ValueError("run_smplify=True needs the GMM pose prior (gmm=)")
    errors = []

    def work(w):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device=device)):
                for r in range(w, len(mine), workers):
                    i, j = mine[r]
                    T = len(dataset["pose"][i])
                    kp_pix = torch.as_tensor(dataset["joint2d_mp"][i][j], dtype=torch.float32) * scale   # evaluate.py:43-44 -> :87
                    p, t, _ = runners[w].run(out_p[r, :T], out_t[r, :T], kp_pix, ori[r, :T], dataset["cam_K"][i][j], lr=0.001)
                    out_p[r, :T], out_t[r, :T] = p, t
                    if smplify_info is not None:
                        smplify_info[(i, j)] = dict(runners[w].last_info)
                torch.cuda.current_stream().synchronize()
        except Exception as e:  # noqa: BLE001 - re-raised on the calling thread
            errors.append(e)

    ths = [threading.Thread(target=work, args=(w,)) for w in range(workers)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    if errors:
        raise errors[0]
    torch.cuda.synchronize()


def joint_errors(model, pose_p, tran_p, pose_t, tran_t):
    """dict of per-sequence means: root-aligned MPJPE, PVE and Procrustes-aligned MPJPE (m) from the fused metric
    kernel (keypoints = the model's regressor joints if ``set_regressor`` was called, else the 24 SMPL joints),
    absolute root position error (m) and global joint rotation error (degrees, float64 atan2 form).
    ``model`` = robustcap_amd.body.ParametricModel."""
    gp, jp = model.forward_kinematics(pose_p, tran=tran_p)
    gt, jt = model.forward_kinematics(pose_t, tran=tran_t)
    _, (mpjpe, pve, pa) = model.mesh_metrics(pose_p, pose_t)
    root = _body.position_error(jp[:, 0], jt[:, 0], device=jp.device)                   # evaluate.py:113-117
    D = (gp.double().transpose(-1, -2) @ gt.double()).reshape(-1, 3, 3)
    v = torch.stack((D[:, 2, 1] - D[:, 1, 2], D[:, 0, 2] - D[:, 2, 0], D[:, 1, 0] - D[:, 0, 1]), dim=1) * 0.5
    ang = torch.rad2deg(torch.atan2(v.norm(dim=1), (D[:, 0, 0] + D[:, 1, 1] + D[:, 2, 2] - 1) * 0.5)).mean()
    return {"mpjpe_smpl24_m": mpjpe, "pve_m": pve, "pa_mpjpe_smpl24_m": pa, "root_error_m": root, "global_angle_deg": float(ang)}


def cal_mpjpe(model, pose, gt_pose, j_regressor=None, cal_pampjpe=False):
    """evaluate.py:120-133: [MPJPE over the first 14 regressor joints (pelvis-aligned), PVE, PA-MPJPE], translation
    zero as in the reference -- one fused kernel per call (rc_mesh_metrics: both meshes skinned in registers, regressor
    dot products and vertex distances accumulated on the fly, per-frame Procrustes on the device).
    ``j_regressor`` [17, V] is the external ``J_regressor_h36m.npy``; without it the 24 SMPL joints stand in for the
    regressor joints (documented deviation: the asset is not shipped)."""
    if j_regressor is not None and getattr(model, "_regressor_id", None) != id(j_regressor):
        model.set_regressor(j_regressor, 14)
    elif j_regressor is None and getattr(model, "_regressor_id", None) is not None:
        raise ValueError("this ParametricModel has a regressor set; pass it again or use a fresh model")
    _, mean = model.mesh_metrics(pose, gt_pose)
    return mean if cal_pampjpe else mean[:2]


def dataset_metrics(model, dataset, results, device="cuda"):
    """``cal_mpjpe`` (evaluate.py:120-133: MPJPE over the regressor joints, PVE, PA-MPJPE, translation zero) of EVERY
    (sequence, camera) row of ``results`` = {(i, j): (pose [T,24,3,3], tran)} in ONE metric call: the reference runs it row by
    row inside its loop (evaluate.py:95-100); here all rows' frames are concatenated, the ground-truth rotations are built
    once per sequence (their root is then turned into each camera's frame, evaluate.py:46-48) and the per-row means are cut
    out of the per-frame result. Returns ({(i, j): [mpjpe, pve, pa_mpjpe]}, their mean over the rows)."""
    dev = torch.device(device)
    rows = list(results)
    if not rows:
        return {}, [float("nan")] * 3
    seq_R, P, G, lens = {}, [], [], []
    for (i, j) in rows:
        if i not in seq_R:
            seq_R[i] = _body.axis_angle_to_rotation_matrix(torch.as_tensor(dataset["pose"][i]).reshape(-1, 3), device).view(-1, 24, 3, 3)
        Rcw = torch.as_tensor(dataset["cam_T"][i][j], dtype=torch.float32)[:3, :3].to(dev)
        gt = seq_R[i].clone()
        gt[:, 0] = Rcw @ gt[:, 0]
        pose = results[(i, j)][0].to(dev)
        n = min(pose.shape[0], gt.shape[0])
        P.append(pose[:n]), G.append(gt[:n]), lens.append(n)
    per_frame, _ = model.mesh_metrics(torch.cat(P), torch.cat(G))
    pf = per_frame.cpu().double()
    out, a = {}, 0
    for (i, j), n in zip(rows, lens):
        out[(i, j)] = [float(v) for v in pf[a:a + n].mean(0)]
        a += n
    return out, [float(v) for v in np.mean(list(out.values()), axis=0)]


def evaluate(dataset, state_dict, body, device="cuda", **kw):
    """Full loop: run all rows, return per-row metrics and their mean (rank-local when not distributed)."""
    res = run_dataset(dataset, state_dict, body, device=device, **kw)
    model = _body.ParametricModel(body=body, device=device)
    per_row = {}
    for (i, j), (pose, tran) in res.items():
        pt, tt = labels(dataset, i, j, device)
        per_row[(i, j)] = joint_errors(model, pose, tran, pt, tt)
    keys = next(iter(per_row.values())).keys() if per_row else []
    mean = {k: float(np.mean([m[k] for m in per_row.values()])) for k in keys}
    return per_row, mean
