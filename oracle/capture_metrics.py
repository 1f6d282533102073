#!/usr/bin/env python3
"""Golden vectors for the evaluation metrics by RUNNING THE REFERENCE here.

TEST INFRASTRUCTURE, like capture_reference.py. Calls the reference's own ``evaluate.cal_mpjpe`` (evaluate.py:120-133:
full-mesh FK, J_regressor, pelvis alignment, MPJPE / PVE / PA-MPJPE via utils.reconstruction_error, utils.py:138-203)
and ``art.PositionErrorEvaluator`` (articulate/evaluator.py:100-129) on seeded poses. External assets are replaced by
seeded stand-ins written into a temp cwd: SMPL pickle (synth.make_body), ``J_regressor_h36m.npy``
(synth.make_j_regressor), ``gmm_08.pkl`` (synth.make_gmm; evaluate.py imports the smplify package).
Writes tests/golden/metrics.npz (numbers only).
"""
import os
import pickle
import sys
import tempfile

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
from robustcap_amd import synth  # noqa: E402
import capture_reference as cr  # noqa: E402
from oracle import _npz  # noqa: E402


def main():
    body = synth.make_body(cr.BODY_SEED)
    tmp = tempfile.mkdtemp(prefix="rc_ref_metrics_")
    cr._write_body_pickle(os.path.join(tmp, "models", "SMPL_male.pkl"), body)
    os.makedirs(os.path.join(tmp, "data", "dataset_work"))
    Jr = synth.make_j_regressor(4)
    np.save(os.path.join(tmp, "data", "dataset_work", "J_regressor_h36m.npy"), Jr)
    with open(os.path.join(tmp, "data", "dataset_work", "gmm_08.pkl"), "wb") as f:
        pickle.dump(synth.make_gmm(3), f)
    os.chdir(tmp)
    cr._install_stubs()
    sys.path.insert(0, cr.REF)
    import articulate as art  # noqa
    import evaluate as ref_eval
    import utils as ref_utils
    t = torch.from_numpy
    g = {}
    T = 24
    gt = t(synth.make_motion(51, 1, T, body, conf="high")["pose"][0].copy())
    near = t(synth._rodrigues((0.08 * synth.normal(52, 0, T * 72)).reshape(T, 24, 3).astype(np.float64)).astype(np.float32))
    pred = gt @ near                                              # a plausible prediction: every joint off by a few degrees
    far = t(synth.make_motion(53, 1, T, body, conf="high")["pose"][0].copy())          # an unrelated motion
    for name, p in (("near", pred), ("far", far), ("same", gt.clone())):
        out = ref_eval.cal_mpjpe(p.clone(), gt.clone(), cal_pampjpe=True)
        g["cal_" + name] = out.numpy().astype(np.float64)
        g["cal2_" + name] = ref_eval.cal_mpjpe(p.clone(), gt.clone()).numpy().astype(np.float64)
        # per-frame pieces, for a sharper comparison than the three means
        _, _, vt = ref_eval.body_model.forward_kinematics(gt.clone(), calc_mesh=True)
        _, _, vp = ref_eval.body_model.forward_kinematics(p.clone(), calc_mesh=True)
        kt, kp = torch.matmul(ref_eval.J_regressor, vt)[:, :14], torch.matmul(ref_eval.J_regressor, vp)[:, :14]
        kt, kp = kt - kt[:, :1], kp - kp[:, :1]
        g["frame_mpjpe_" + name] = (kt - kp).norm(dim=2).mean(dim=1).numpy()
        g["frame_pve_" + name] = (vt - vp).norm(dim=2).mean(dim=1).numpy()
        g["frame_pa_" + name] = ref_utils.reconstruction_error(kp.numpy(), kt.numpy(), reduction=None)
    g["pose_gt"], g["pose_near"], g["pose_far"] = gt.numpy(), pred.numpy(), far.numpy()
    # Procrustes on raw point sets, including a mirrored one (det < 0 branch of utils.py:171-174)
    S2 = synth.normal(54, 0, 6 * 14 * 3).reshape(6, 14, 3).astype(np.float32)
    A = synth._rodrigues(synth.normal(54, 1, 6 * 3).reshape(6, 3).astype(np.float64)).astype(np.float32)
    S1 = 1.7 * np.einsum("tij,tkj->tki", A, S2) + 0.3 + 0.05 * synth.normal(54, 2, 6 * 14 * 3).reshape(6, 14, 3)
    S1[3:] *= np.array([1.0, 1.0, -1.0], np.float32)              # reflections: the best ROTATION is not the reflection
    S1 = S1.astype(np.float32)
    g["pa_S1"], g["pa_S2"] = S1, S2
    g["pa_err"] = ref_utils.reconstruction_error(S1, S2, reduction=None)
    g["pa_hat"] = ref_utils.compute_similarity_transform_batch(S1, S2)
    # root position error (evaluate.py:113-117)
    a, b = synth.normal(55, 0, 90).reshape(30, 3), synth.normal(55, 1, 90).reshape(30, 3)
    g["pos_a"], g["pos_b"] = a, b
    g["pos_err"] = np.float64(art.PositionErrorEvaluator()(t(a), t(b)))
    _npz.save(os.path.join(cr.OUT, "metrics.npz"), **g)
    cr.write_hashes()
    for k in ("cal_near", "cal_far", "cal_same", "pa_err", "pos_err"):
        print(k, g[k])


if __name__ == "__main__":
    main()
