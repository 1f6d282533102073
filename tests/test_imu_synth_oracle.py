"""Pin the IMU-synthesis oracle to the reference capture (oracle/capture_imu_synth.py). CPU only."""
import os

import numpy as np
import pytest

from oracle import imu_synth_oracle as I


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "imu_synth.npz"))


@pytest.mark.parametrize("name,n", [("long", 2), ("long", 1), ("five", 2), ("five", 1), ("four", 1), ("three", 1)])
def test_syn_acc_bit_exact(name, n, g):
    assert np.array_equal(I.syn_acc(g["v_" + name], n), g["acc%d_%s" % (n, name)])


def test_syn_acc_rejects_what_the_reference_rejects(g):
    with pytest.raises(ValueError):
        I.syn_acc(g["v_four"], 2)


def test_recipe_matches_reference(g, synth_assets):
    ori, acc, joint, vert = I.synthesize_imu(synth_assets["body"], g["pose_aa"], g["tran"])
    assert np.abs(ori - g["imu_ori"]).max() <= 1e-6
    assert np.abs(joint - g["joint3d"]).max() <= 2e-6 and np.abs(vert - g["vert6"]).max() <= 2e-6
    assert np.array_equal(I.syn_acc(g["vert6"]), g["imu_acc"])                        # the stencil itself: exact
    assert np.abs(acc - g["imu_acc"]).max() <= 3600 * 4 * 2e-6                        # 1e-6 m of vertex noise x the stencil gain
