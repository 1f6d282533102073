"""IMU synthesis on the GPU against the reference capture tests/golden/imu_synth.npz."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
t = torch.from_numpy


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "imu_synth.npz"))


@pytest.mark.parametrize("name,n", [("long", 2), ("long", 1), ("five", 2), ("five", 1), ("four", 1), ("three", 1)])
def test_syn_acc_bit_exact(name, n, g):
    from robustcap_amd.preprocess import _syn_acc
    out = _syn_acc(t(g["v_" + name]), smooth_n=n)
    assert out.shape == g["v_" + name].shape
    assert np.array_equal(out.cpu().numpy(), g["acc%d_%s" % (n, name)])


def test_syn_acc_edges(g):
    from robustcap_amd import _lib
    from robustcap_amd.preprocess import _syn_acc
    with pytest.raises(_lib.RobustcapLibraryError):                 # the reference raises below 2n+1 frames
        _syn_acc(t(g["v_four"]), smooth_n=2)
    assert _syn_acc(torch.zeros(0, 6, 3)).shape == (0, 6, 3)
    flat = _syn_acc(t(g["v_long"]).reshape(40, 18), smooth_n=2)     # any trailing shape
    assert np.array_equal(flat.cpu().numpy().reshape(40, 6, 3), g["acc2_long"])


def test_recipe_matches_reference(g, synth_assets):
    from robustcap_amd.body import ParametricModel
    from robustcap_amd.preprocess import synthesize_imu
    model = ParametricModel(body=synth_assets["body"])
    ori, acc, joint, vert6 = synthesize_imu(model, t(g["pose_aa"]), t(g["tran"]))
    assert float((ori.cpu() - t(g["imu_ori"])).abs().max()) <= 2e-6
    assert float((joint.cpu() - t(g["joint3d"])).abs().max()) <= 2e-6
    assert float((vert6.cpu() - t(g["vert6"])).abs().max()) <= 2e-6
    assert float((acc.cpu() - t(g["imu_acc"])).abs().max()) <= 3600 * 4 * 2e-6          # vertex noise x stencil gain
    R = t(g["pose_aa"])
    from robustcap_amd.body import axis_angle_to_rotation_matrix
    o2, a2, _, _ = synthesize_imu(model, axis_angle_to_rotation_matrix(R.reshape(-1, 3)).view(-1, 24, 3, 3), t(g["tran"]))
    assert torch.equal(o2, ori) and torch.equal(a2, acc)                                  # rotation-matrix input: same path
    full = model.forward_mesh(axis_angle_to_rotation_matrix(R.reshape(-1, 3)).view(-1, 24, 3, 3), t(g["tran"]))
    from robustcap_amd import config as C
    assert float((full[:, list(C.vi_mask)] - vert6).abs().max()) <= 1e-6                  # the six vertices of the full sweep
