"""smplify optimiser timing on one synthetic sequence (frames x 26 closure evaluations), product path only.
usage: python tools/smplify_bench.py [T]
(The CPU-oracle comparison of the same workload lives in tests/test_gpu_smplify.py::test_long_sequence_improves.)"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from robustcap_amd import config as C           # noqa: E402
from robustcap_amd import synth                 # noqa: E402
from robustcap_amd.body import axis_angle_to_rotation_matrix  # noqa: E402
from robustcap_amd.smplify import TemporalSMPLify  # noqa: E402

t = torch.from_numpy


def make_case(runner, body, T, seed=11):
    """ground-truth motion -> pixel keypoints by the product's own FK; the 'prediction' is the truth with every joint
    rotated by ~3 degrees and the root shifted by 2 cm."""
    m = synth.make_motion(seed, 1, T, body, conf="high")
    K = torch.tensor([[600.0, 0.0, 320.0], [0.0, 600.0, 240.0], [0.0, 0.0, 1.0]])
    pose, tran = t(np.asarray(m["pose"][0], np.float32)), t(np.asarray(m["tran"][0], np.float32))
    _, _, mj = runner.model.forward_kinematics(pose, tran=tran, calc_mesh=True)
    proj = (K.to(mj.device) @ (mj / mj[..., 2:]).unsqueeze(-1)).squeeze(-1)[..., :2]
    kp = torch.cat([proj, torch.full((T, 33, 1), 0.9, device=mj.device)], dim=-1)
    noisy = axis_angle_to_rotation_matrix(0.05 * t(synth.normal(seed + 1, 0, T * 72).reshape(-1, 3))).view(T, 24, 3, 3)
    pose0 = pose.to(mj.device) @ noisy
    gp, _ = runner.model.forward_kinematics(pose0, tran=tran)
    return pose0, tran + 0.02, kp, gp[:, list(C.ji_mask)].contiguous(), K


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 600
    body, gmm = synth.make_body(1), synth.make_gmm(3)
    runner = TemporalSMPLify(body=body, gmm=gmm)
    case = make_case(runner, body, T)
    res = lambda p, q: float(runner.get_fitting_loss(p, q, case[2], case[4]).mean())
    runner.run(*case, lr=0.001)                                    # warm-up: allocations, code load
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pose, tran, update = runner.run(*case, lr=0.001)
    torch.cuda.synchronize()
    out = {"frames": T, "hip": dict(runner.last_info, wall_ms=1e3 * (time.perf_counter() - t0), updated=int(update.sum())),
           "residual": {"before": res(case[0], case[1]), "after": res(pose, tran)}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
