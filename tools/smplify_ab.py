#!/usr/bin/env python3
"""A/B of the smplify closure between two builds of the library (RC_LIB_PATH selects the build): one evaluation of loss and
gradient on a seeded 600-frame row, written to an .npz; with two files given, compares them element by element.
    python tools/smplify_ab.py gpurun_out/ab_new.npz
    RC_LIB_PATH=.../other.so python tools/smplify_ab.py gpurun_out/ab_old.npz
    python tools/smplify_ab.py gpurun_out/ab_new.npz gpurun_out/ab_old.npz"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def compare(a, b):
    A, B = np.load(a), np.load(b)
    for k in A.files:
        x, y = A[k], B[k]
        ne = int((x != y).sum())
        print(f"{k}: shape {x.shape} differing {ne} max|d| {float(np.abs(x - y).max()):.3e} max|x| {float(np.abs(x).max()):.3e}")
        if k == "grad_pose" and ne:
            print("  components with differences:", sorted(set(np.nonzero(x != y)[1].tolist()))[:80])


def main():
    if len(sys.argv) > 2:
        return compare(sys.argv[1], sys.argv[2])
    import torch
    from robustcap_amd import synth
    from robustcap_amd.smplify import TemporalSMPLify
    T, seed = 600, 11
    body, gmm = synth.make_body(1), synth.make_gmm(3)
    runner = TemporalSMPLify(body=body, gmm=gmm)
    rnd = lambda stream, *shape: synth.normal(seed, stream, int(np.prod(shape))).reshape(shape).astype(np.float32)
    t = torch.from_numpy
    bp = t(0.35 * rnd(0, T, 72))
    tr = t((np.array([0.1, -0.2, 3.0], np.float32) + 0.2 * rnd(1, T, 3)).astype(np.float32))
    K = torch.tensor([[600.0, 0.0, 320.0], [0.0, 600.0, 240.0], [0.0, 0.0, 1.0]])
    kp = torch.cat([t(np.array([320.0, 240.0], np.float32) + 120.0 * rnd(2, T, 33, 2)), t(synth.uniform01(seed, 3, T * 33).reshape(T, 33, 1).astype(np.float32))], dim=-1)
    ref3d = t(np.array([0.1, -0.2, 3.0], np.float32) + 0.4 * rnd(4, T, 33, 3))
    imu_aa = t(0.5 * rnd(5, T, 18))
    loss, gp, gt = runner.loss_and_grad(bp, tr, kp, ref3d, imu_aa, K)
    np.savez(sys.argv[1], loss=np.float64(loss), grad_pose=gp.cpu().numpy(), grad_tran=gt.cpu().numpy())
    print("loss", loss)


if __name__ == "__main__":
    main()
