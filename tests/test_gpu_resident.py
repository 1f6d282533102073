"""Resident layer-step kernel of the wavefront engine (rc_gemm_lds.hip: rc_gemm_resident_kernel; rc_api.cpp: run_wave2_segment, `resident`).

north_star asks for "a fused persistent kernel ... hidden state kept ... across timesteps"; the reference's whole-sequence form is
articulate/utils/torch/rnn.py:129-133 and its frame loop net/sig_mp.py:113-274. One launch carries the LSTM layer steps and linear1 layers
of every tick of a call; what stream order and events do in the stream engine, counters in device memory do here. Its results are bitwise
those of the frame-stepped launches -- outputs, final states, traces -- and within the reference's tolerance of the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from robustcap_amd import synth
from robustcap_amd.net.sig_mp import Net

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NETS = ("rnn2", "rnn3", "rnn4", "rnn6", "rnn7", "rnn8")


def _run(sd, body, m, B, resident, seq=True, cut=None):
    t = torch.from_numpy
    net = Net(body=body, batch=B)
    net.load_state_dict(sd)
    net.gravityc = t(m["gravityc"])
    net.set_sequence_mode(seq, 8, force=True)
    net.set_resident(resident)
    T = m["j2dc"].shape[1]
    P, Tr = [], []
    for lo, hi in ((0, cut or T), (cut or T, T)):
        if hi > lo:
            p, tr = net.forward_sequence(t(m["j2dc"][:, lo:hi]), t(m["accc"][:, lo:hi]), t(m["oric"][:, lo:hi]), first_frame=(lo == 0))
            P.append(p), Tr.append(tr)
    torch.cuda.synchronize()
    st = [net.get_state(n) for n in NETS]
    return torch.cat(P, 1).cpu(), torch.cat(Tr, 1).cpu(), st, net.get_trace(), net.sequence_stats(), net.resident_stats()


def _schedule(rng, B, T):
    """per-row confidences in runs of 1..9 frames: occluded / mid / visible, some rows on the thresholds (riders, init_net waits, lagging rows)"""
    c = np.empty((B, T), np.float32)
    for b in range(B):
        i = 0
        while i < T:
            n = int(rng.integers(1, 10))
            r = rng.random()
            v = rng.uniform(0.3, 0.69) if r < 0.35 else (rng.uniform(0.71, 0.79) if r < 0.55 else rng.uniform(0.81, 0.99))
            if rng.random() < 0.05:
                v = float(rng.choice([0.7, 0.8, 0.69999, 0.70001]))
            c[b, i:i + n] = v
            i += n
    return c


@pytest.mark.parametrize("B,T,conf,cut", [(256, 40, "high", None), (256, 48, "switching", 19), (192, 36, "switching", None), (160, 30, "mixed", 11)])
def test_resident_engine_is_bitwise_the_frame_stepped_path(B, T, conf, cut):
    sd, body = synth.make_state_dict(0), synth.make_body(1)
    m = synth.make_motion(70 + B, 16, T, body, conf="high" if conf == "switching" else conf)
    rep = (B + 15) // 16
    m = {k: np.concatenate([v] * rep, 0)[:B].copy() for k, v in m.items()}
    if conf == "switching":
        m["j2dc"][..., 2] = _schedule(np.random.default_rng(B + T), B, T)[:, :, None]
    rp, rt, rs, rtr, rstat, rres = _run(sd, body, m, B, True, cut=cut)
    sp, st_, ss, strc, sstat, _ = _run(sd, body, m, B, False, seq=False, cut=cut)
    assert rres[0] >= 1 and rres[1] == 0, rres                       # the resident kernel ran, no wait ran out
    assert rstat[0] > 0 and sstat[0] == 0
    assert torch.equal(rp, sp) and torch.equal(rt, st_) and torch.equal(rtr, strc)
    for (h1, c1), (h2, c2) in zip(rs, ss):
        assert torch.equal(h1, h2) and torch.equal(c1, c2)
    assert bool(torch.isfinite(rp).all())


def test_resident_engine_against_the_oracle():
    """<= 1e-4 m / 0.1 deg of the CPU restatement (north_star's bound) on a batch with occluded stretches."""
    from oracle import sig_mp_oracle as O          # checker only
    B, T = 160, 20
    sd, body = synth.make_state_dict(0), synth.make_body(1)
    m = synth.make_motion(9, B, T, body, conf="mixed")
    p, tr, *_rest, res = _run(sd, body, m, B, True)
    assert res[0] >= 1 and res[1] == 0
    t = torch.from_numpy
    ref = O.OracleNet(body, batch=B)
    ref.load_numpy_state_dict(sd)
    ref.gravityc = t(m["gravityc"])
    for i in range(T):
        rp, rtr = ref.forward_batch(t(m["j2dc"][:, i]), t(m["accc"][:, i]), t(m["oric"][:, i]), None, i == 0)
        assert float((p[:, i] - rp).abs().max()) < 1e-4, i             # rotation-matrix entries: 1e-4 ~ 0.006 deg
        assert float((tr[:, i] - rtr).abs().max()) < 1e-4, i


def test_linear1_items_of_the_resident_kernel_equal_the_wide_tile_launch():
    """RC_DBG_DENSE_ITEMS=1 (rc_api.cpp: dense_items_selftest): every linear1 launch of a frame-stepped run is also computed as one tick
    of the resident kernel; the two must agree bit for bit on every element the launch writes (incl. K' = 128, riders' inputs)."""
    code = ("import sys, torch; sys.path.insert(0, %r)\n"
            "from robustcap_amd import synth\nfrom robustcap_amd.net.sig_mp import Net\n"
            "sd, body = synth.make_state_dict(0), synth.make_body(1)\nB, T = 256, 4\n"
            "m = synth.make_motion(3, B, T, body, conf='mixed')\nt = torch.from_numpy\n"
            "net = Net(body=body, batch=B); net.load_state_dict(sd); net.gravityc = t(m['gravityc'])\n"
            "net.set_sequence_mode(False, 8, force=True)\n"
            "net.forward_sequence(t(m['j2dc']), t(m['accc']), t(m['oric']), first_frame=True); torch.cuda.synchronize()\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, RC_DBG_DENSE_ITEMS="1"), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stderr.splitlines() if ln.startswith("[dbg dense]")]
    assert len(lines) >= 8, r.stderr[-2000:]
    checked = 0
    for ln in lines:
        bad, written = int(ln.split(":")[1].split()[0]), int(ln.split(" of ")[1].split()[0])
        assert bad == 0, ln
        checked += written
    assert checked > 1_000_000
