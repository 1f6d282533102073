"""The benchmark's input generator on the device (robustcap_amd.preprocess.make_motion_device; SURVEY.md 8(f) rank 3: "on-device generator for
benchmark inputs (FK -> ori / acc / 2D)"). Recipe: preprocess.py:22-33 (`_syn_acc`), :206-222 (IMU orientations = global rotations of joints
ji_mask, accelerations of vertices vi_mask) and the projection of evaluate.py:70-72. The device generator must reproduce the host numpy
generator (`synth.make_motion`, float64 inside) to fp32 rounding: same seeds, same trajectories."""
import numpy as np
import pytest
import torch

import bench
from robustcap_amd import preprocess, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("conf,T", [("mixed", 40), ("high", 7), ("occ", 24)])
def test_device_generator_equals_the_host_generator(conf, T):
    body = synth.make_body(1)
    host = synth.make_motion(3, 5, T, body, conf=conf)
    dev = preprocess.make_motion_device(3, 5, T, body, conf=conf)
    for k, tol in (("j2dc", 2e-5), ("oric", 2e-5), ("tran", 1e-6), ("pose", 1e-6)):
        d = float((dev[k].cpu() - torch.from_numpy(host[k])).abs().max())
        assert d <= tol, (k, d)
    # accelerations are second differences * 3600 of metre-scale positions: fp32 positions carry ~3e-7 m, i.e. ~4e-3 m/s^2
    da = float((dev["accc"].cpu() - torch.from_numpy(host["accc"])).abs().max())
    assert da <= 2e-2 and da <= 1e-3 * float(np.abs(host["accc"]).max()) + 2e-2, da
    assert np.array_equal(dev["gravityc"], host["gravityc"]) and np.array_equal(dev["first_tran"], host["first_tran"])
    assert np.allclose(dev["conf"], host["conf"], atol=1e-6)


def test_bench_inputs_are_born_on_the_device_and_match_the_host_recipe():
    body = synth.make_body(1)
    h = bench.make_inputs(body, 40, 20, "mixed", seed=2, unique=8)
    d = bench.make_inputs_device(body, 40, 20, "mixed", seed=2, unique=8)
    assert d["j2dc"].is_cuda and d["accc"].is_cuda and d["oric"].is_cuda
    assert float((d["j2dc"].cpu() - torch.from_numpy(h["j2dc"])).abs().max()) <= 2e-5      # incl. the per-body confidence schedules of the tiled copies
    assert float((d["oric"].cpu() - torch.from_numpy(h["oric"])).abs().max()) <= 2e-5
    assert np.allclose(d["conf"], h["conf"], atol=1e-5)
