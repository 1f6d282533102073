"""Pin the metrics oracle to the reference capture (oracle/capture_metrics.py). CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import metrics_oracle as M
from robustcap_amd import synth

t = torch.from_numpy


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "metrics.npz"))


def test_regressor_generator_is_convex():
    Jr = synth.make_j_regressor(4)
    assert Jr.shape == (17, 6890) and (Jr >= 0).all() and np.allclose(Jr.sum(1), 1.0, atol=1e-6)
    assert np.array_equal(Jr, synth.make_j_regressor(4))


@pytest.mark.parametrize("name", ["near", "far", "same"])
def test_cal_mpjpe_matches_reference(name, g, synth_assets):
    mp, pve, pa = M.frame_metrics(synth_assets["body"], synth.make_j_regressor(4), t(g["pose_" + name] if name != "same" else g["pose_gt"]),
                                  t(g["pose_gt"]))
    assert np.abs(mp - g["frame_mpjpe_" + name]).max() <= 2e-6
    assert np.abs(pve - g["frame_pve_" + name]).max() <= 2e-6
    assert np.abs(pa - g["frame_pa_" + name]).max() <= 2e-6
    assert np.allclose([mp.mean(), pve.mean(), pa.mean()], g["cal_" + name], atol=2e-6)
    assert np.allclose(g["cal2_" + name], g["cal_" + name][:2])


def test_procrustes_matches_reference(g):
    err = M.reconstruction_error(g["pa_S1"], g["pa_S2"])
    assert np.abs(err - g["pa_err"]).max() <= 1e-5 * np.abs(g["pa_err"]).max()
    hat = np.stack([M.similarity_transform(a, b) for a, b in zip(g["pa_S1"], g["pa_S2"])])
    assert np.abs(hat - g["pa_hat"]).max() <= 1e-5
    assert g["pa_err"][3:].min() > 10 * g["pa_err"][:3].max()             # reflections cannot be rotated away


def test_position_error_matches_reference(g):
    assert abs(M.position_error(g["pos_a"], g["pos_b"]) - float(g["pos_err"])) <= 1e-6
