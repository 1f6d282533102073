"""smplify optimiser: device path against the CPU oracle on one synthetic sequence (frames x 26 closure evaluations).
usage: python tools/smplify_bench.py [T] [--oracle]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import sig_mp_oracle as O          # noqa: E402  (checker / CPU baseline only)
from oracle import smplify_oracle as S          # noqa: E402
from robustcap_amd import config as C           # noqa: E402
from robustcap_amd import synth                 # noqa: E402
from robustcap_amd.smplify import TemporalSMPLify  # noqa: E402

t = torch.from_numpy


def make_case(body, T, seed=11):
    m = synth.make_motion(seed, 1, T, body, conf="high")
    K = torch.tensor([[600.0, 0.0, 320.0], [0.0, 600.0, 240.0], [0.0, 0.0, 1.0]])
    obody = O.OracleBody(body)
    pose, tran = t(np.asarray(m["pose"][0], np.float32)), t(np.asarray(m["tran"][0], np.float32))
    with torch.no_grad():
        _, joint, vert = obody.forward_kinematics(pose, tran)
        mj = obody.landmarks(vert, joint)
        proj = (K @ (mj / mj[..., 2:]).unsqueeze(-1)).squeeze(-1)[..., :2]
        kp = torch.cat([proj, torch.full((T, 33, 1), 0.9)], dim=-1)
        noisy = O.axis_angle_to_rotation_matrix(0.05 * t(synth.normal(seed + 1, 0, T * 72).reshape(-1, 3))).view(T, 24, 3, 3)
        pose0 = pose @ noisy
        gp, _, _ = obody.forward_kinematics(pose0, tran)
    return pose0, tran + 0.02, kp, gp[:, list(C.ji_mask)].contiguous(), K


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 600
    body, gmm = synth.make_body(1), synth.make_gmm(3)
    case = make_case(body, T)
    out = {"frames": T}
    runner = TemporalSMPLify(body=body, gmm=gmm)
    runner.run(*case, lr=0.001)                                    # warm-up: allocations, code load
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pose, tran, update = runner.run(*case, lr=0.001)
    torch.cuda.synchronize()
    out["hip"] = dict(runner.last_info, wall_ms=1e3 * (time.perf_counter() - t0), updated=int(update.sum()))
    if "--oracle" in sys.argv:
        t0 = time.perf_counter()
        po, to, uo = S.smplify_runner(body, gmm, *case[:2], case[2], case[3], case[4])
        out["oracle"] = {"wall_ms": 1e3 * (time.perf_counter() - t0), "updated": int(uo.sum()), "threads": torch.get_num_threads()}
        ob = O.OracleBody(body)
        res = lambda p, q: float(O.reprojection_residual(ob, p, q, case[2], case[4]).mean())
        out["residual"] = {"before": res(case[0], case[1]), "hip": res(pose.cpu(), tran.cpu()), "oracle": res(po, to)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
