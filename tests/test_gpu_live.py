"""The lean live frame (csrc/rc_live.hip; BASELINE config 5) through the C ABI: the seven-launch capture of the steady-state
frame against the REFERENCE's captured sequences, against the frame-stepped plan (rc_step), and graph replay == direct launches."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SEQS = sorted(glob.glob(os.path.join(GOLD, "seq_*.npz")))
t = torch.from_numpy


def maxdiff(a, b):
    return float((torch.as_tensor(a).cpu().double() - torch.as_tensor(b).cpu().double()).abs().max())


def make_net(synth_assets, batch=1, live_ctor=False):
    """live_ctor mirrors `Net.live = True` BEFORE construction (evaluate.py:392); like the reference, `live` stays a class
    attribute that the instance reads at call time -- reset when the test is over."""
    from robustcap_amd.net.sig_mp import Net
    Net.live = live_ctor
    net = Net(body=synth_assets["body"], batch=batch)
    net.load_state_dict(synth_assets["state_dict"])
    return net


@pytest.fixture(autouse=True)
def _reset_class_live():
    yield
    from robustcap_amd.net.sig_mp import Net
    Net.live = False


@pytest.mark.parametrize("path", SEQS, ids=[os.path.basename(p)[4:-4] for p in SEQS])
def test_live_capture_vs_reference_sequences(path, synth_assets):
    """Every reference sequence through forward_online with use_graph (rc_live_step): <= 1e-4 m / 0.1 deg against the REFERENCE's
    own outputs, exact branch traces, final LSTM states -- with the steady-state frames on the lean seven-launch capture."""
    from oracle import sig_mp_oracle as O
    s = np.load(path)
    live = str(s["live"])
    net = make_net(synth_assets, 1, live_ctor=(live == "pre"))
    if live == "post":
        net.live = True
    net.use_flat_floor = bool(s["use_flat_floor"])
    net.use_reproj_opt = bool(s["use_reproj_opt"]) if "use_reproj_opt" in s else False
    net.use_vision_updater = bool(s["use_vision_updater"]) if "use_vision_updater" in s else True
    net.use_imu_updater = bool(s["use_imu_updater"]) if "use_imu_updater" in s else True
    net.gravityc = t(s["gravityc"])
    net.use_graph = True
    ft = t(s["first_tran"]) if s["first_tran"].size else None
    T = s["pose"].shape[0]
    poses, trans = [], []
    for i in range(T):
        p, tr = net.forward_online(t(s["j2dc"][i]), t(s["accc"][i]), t(s["oric"][i]), ft if i == 0 else None,
                                   bool(s["first_frame"]) and i == 0)
        tc = net.get_trace()[0].tolist()
        exp = s["trace"][i]
        assert tc[1] == int(exp[1]) and tc[2] == int(exp[2]), f"frame {i}: rnn4/rnn6 step counts {tc} vs {exp}"
        assert tc[3] == int(exp[4]) and tc[4] == int(exp[5]), f"frame {i}: floor/reach {tc} vs {exp}"
        poses.append(p.clone()), trans.append(tr.clone())
    pose, tran = torch.stack(poses), torch.stack(trans)
    rp, rt = t(s["pose"]), t(s["tran"])
    assert maxdiff(tran, rt) <= 1e-4
    assert float(O.rotation_angle_deg(pose, rp).max()) <= 0.1
    ob = O.OracleBody(synth_assets["body"])
    jp = ob.forward_kinematics(pose.reshape(-1, 24, 3, 3), tran.reshape(-1, 3))[1]
    jr = ob.forward_kinematics(rp.reshape(-1, 24, 3, 3), rt.reshape(-1, 3))[1]
    assert maxdiff(jp, jr) <= 1e-4
    for n in ("rnn2", "rnn3", "rnn4", "rnn6", "rnn7", "rnn8"):
        h, c = net.get_state(n)
        assert maxdiff(h[:, 0], s["h_" + n]) <= 1e-4 and maxdiff(c[:, 0], s["c_" + n]) <= 2e-4, n
    lean, full = net.live_stats()
    assert lean + full == T
    if T >= 64:
        assert lean >= T // 2, (lean, full)            # the steady state IS the lean capture


def _mixed_motion(synth_assets, seed, T):
    from robustcap_amd import synth
    m = synth.make_motion(seed, 1, T, synth_assets["body"], conf="mixed")
    m["j2dc"][0, 10:25, :, 2] = 0.4                              # an occluded stretch: the deferred updater rides the lean frames
    for i, c in zip(range(28, 36), (0.7, 0.70001, 0.69999, 0.7, 0.9, 0.69995, 0.70005, 0.5)):
        m["j2dc"][0, i, :, 2] = c                                # hugging conf_lo: the host-side choice of capture must stay safe
    for i, c in zip(range(40, 46), (0.8, 0.80001, 0.79999, 0.8, 0.79995, 0.80005)):
        m["j2dc"][0, i, :, 2] = c                                # and conf_hi (the init_net trigger, L178-183)
    return m


def test_lean_frames_equal_the_frame_stepped_plan_to_rounding(synth_assets):
    """forward_online via rc_live_step (lean capture where it applies) against rc_step on the same frames: same branch traces and
    the outputs equal to rounding -- the layer steps are the same MFMA chains, linear2 sums per-tile partial products."""
    T = 120
    m = _mixed_motion(synth_assets, 95, T)
    a, b = make_net(synth_assets, 1), make_net(synth_assets, 1)
    a.gravityc = b.gravityc = t(m["gravityc"])
    b.use_graph = True
    worst = 0.0
    for i in range(T):
        args = (t(m["j2dc"][0, i]), t(m["accc"][0, i]), t(m["oric"][0, i]))
        pa, ta = a.forward_online(*args, first_frame=(i == 0))
        pb, tb = b.forward_online(*args, first_frame=(i == 0))
        assert a.get_trace()[0].tolist() == b.get_trace()[0].tolist(), i
        worst = max(worst, maxdiff(pa, pb), maxdiff(ta, tb))
    assert worst <= 1e-5, worst
    lean, full = b.live_stats()
    assert lean >= T // 2 and full >= 3, (lean, full)          # first frame, init_net frame, transitions on the full captures
    for n in ("rnn2", "rnn3", "rnn4", "rnn6", "rnn7", "rnn8"):
        (ha, ca), (hb, cb) = a.get_state(n), b.get_state(n)
        assert maxdiff(ha, hb) <= 1e-5 and maxdiff(ca, cb) <= 1e-5, n


def test_lean_graph_replay_equals_direct_launches(synth_assets, monkeypatch):
    """config 5: the hipGraph replay of the live frame == the same kernels launched directly (RC_LIVE_EAGER), bitwise; and the
    switch RC_LIVE_LEAN=0 (every frame on the frame-stepped captures) == rc_step bitwise."""
    T = 60
    m = _mixed_motion(synth_assets, 96, T)
    g = make_net(synth_assets, 1)
    monkeypatch.setenv("RC_LIVE_EAGER", "1")
    e = make_net(synth_assets, 1)
    monkeypatch.delenv("RC_LIVE_EAGER")
    monkeypatch.setenv("RC_LIVE_LEAN", "0")
    f = make_net(synth_assets, 1)
    monkeypatch.delenv("RC_LIVE_LEAN")
    s = make_net(synth_assets, 1)
    for n in (g, e, f, s):
        n.gravityc = t(m["gravityc"])
    g.use_graph = e.use_graph = f.use_graph = True
    for i in range(T):
        args = (t(m["j2dc"][0, i]), t(m["accc"][0, i]), t(m["oric"][0, i]))
        pg, tg = g.forward_online(*args, first_frame=(i == 0))
        pe, te = e.forward_online(*args, first_frame=(i == 0))
        pf, tf = f.forward_online(*args, first_frame=(i == 0))
        ps, ts = s.forward_online(*args, first_frame=(i == 0))
        assert torch.equal(pg, pe) and torch.equal(tg, te), i
        assert torch.equal(pf, ps) and torch.equal(tf, ts), i
    assert g.live_stats() == e.live_stats() and g.live_stats()[0] > 0
    assert f.live_stats()[0] == 0


def test_lean_frame_batch_4_rows_in_different_regimes(synth_assets):
    """Four rows (the lean plan's maximum) in different regimes per frame, against forward_batch (frame-stepped) on the same
    inputs: traces equal, outputs and states to rounding; rows are independent in the lean kernels too."""
    from robustcap_amd import synth
    B, T = 4, 90
    m = synth.make_motion(141, B, T, synth_assets["body"], conf="mixed")
    m["j2dc"][1, 20:50, :, 2] = 0.45                             # row 1 occluded while the others see the camera
    m["j2dc"][2, :, :, 2] = 0.95                                 # row 2 always high
    m["j2dc"][3, 5:, :, 2] = 0.3                                 # row 3 occluded for good after frame 5
    a, b = make_net(synth_assets, B), make_net(synth_assets, B)
    a.gravityc = b.gravityc = t(m["gravityc"])
    worst = 0.0
    for i in range(T):
        args = (t(m["j2dc"][:, i]), t(m["accc"][:, i]), t(m["oric"][:, i]))
        pa, ta = a.forward_batch(*args, first_frame=(i == 0))
        pb, tb = b.forward_live(*args, first_frame=(i == 0))
        assert a.get_trace().tolist() == b.get_trace().tolist(), i
        worst = max(worst, maxdiff(pa, pb), maxdiff(ta, tb))
    assert worst <= 1e-5, worst
    lean, full = b.live_stats()
    assert lean >= T // 2, (lean, full)
    for n in ("rnn4", "rnn6", "rnn7"):
        (ha, ca), (hb, cb) = a.get_state(n), b.get_state(n)
        assert maxdiff(ha, hb) <= 1e-5 and maxdiff(ca, cb) <= 1e-5, n
    # a single row of the batch run alone gives the same bits (no cross-row term, no batch-dependent order)
    c = make_net(synth_assets, 1)
    c.gravityc = t(m["gravityc"][2:3])
    d = make_net(synth_assets, B)
    d.gravityc = t(m["gravityc"])
    for i in range(30):
        pc, tc = c.forward_live(t(m["j2dc"][2:3, i]), t(m["accc"][2:3, i]), t(m["oric"][2:3, i]), first_frame=(i == 0))
        pd, td = d.forward_live(t(m["j2dc"][:, i]), t(m["accc"][:, i]), t(m["oric"][:, i]), first_frame=(i == 0))
        assert torch.equal(pc[0], pd[2]) and torch.equal(tc[0], td[2]), i


@pytest.mark.parametrize("aql", ["1", "0"], ids=["aql", "graph_replay"])
def test_offplan_frames_are_caught_on_the_device_and_replayed(synth_assets, monkeypatch, aql):
    """The lean plan checks ITSELF that a frame is its own (rc_live_k1: no row needs a transition step, no row triggers init_net); the
    host-side mirror of those flags in rc_live_step only keeps such frames away beforehand. With the mirror blinded
    (RC_LIVE_MIRROR_BLIND=1) every steady frame is offered to the lean plan: the frames it rejects must change nothing and come back from
    the full capture -- same branch traces as rc_step, outputs and states to rounding, thresholds hugged (conf_lo, conf_hi +- 1e-5) -- on
    the AQL chain and on the graph replay, one row and four rows in different regimes. Without the blindfold nothing is ever replayed."""
    from robustcap_amd import synth
    monkeypatch.setenv("RC_LIVE_AQL", aql)
    T = 120
    m = _mixed_motion(synth_assets, 97, T)
    for blind in ("1", "0"):
        monkeypatch.setenv("RC_LIVE_MIRROR_BLIND", blind)
        a, b = make_net(synth_assets, 1), make_net(synth_assets, 1)
        a.gravityc = b.gravityc = t(m["gravityc"])
        b.use_graph = True
        worst = 0.0
        for i in range(T):
            args = (t(m["j2dc"][0, i]), t(m["accc"][0, i]), t(m["oric"][0, i]))
            pa, ta = a.forward_online(*args, first_frame=(i == 0))
            pb, tb = b.forward_online(*args, first_frame=(i == 0))
            assert a.get_trace()[0].tolist() == b.get_trace()[0].tolist(), (blind, i)
            worst = max(worst, maxdiff(pa, pb), maxdiff(ta, tb))
        assert worst <= 1e-5, (blind, worst)
        for n in ("rnn2", "rnn3", "rnn4", "rnn6", "rnn7", "rnn8"):
            (ha, ca), (hb, cb) = a.get_state(n), b.get_state(n)
            assert maxdiff(ha, hb) <= 1e-5 and maxdiff(ca, cb) <= 1e-5, (blind, n)
        lean, full = b.live_stats()
        assert lean + full == T
        if blind == "1":
            assert b.live_replayed() >= 2 and full >= 3, (b.live_replayed(), lean, full)   # the init_net frame and the transitions
        else:
            assert b.live_replayed() == 0
        del a, b
    # four rows: one row off the plan takes the whole frame to the full capture -- here with the next frame queued ahead of every lean frame
    # (RC_LIVE_SPIN=2), so the frames that are off the plan are caught by a first kernel that was already waiting for them
    monkeypatch.setenv("RC_LIVE_MIRROR_BLIND", "1")
    monkeypatch.setenv("RC_LIVE_SPIN", "2")
    B, T = 4, 90
    m = synth.make_motion(143, B, T, synth_assets["body"], conf="mixed")
    m["j2dc"][1, 20:50, :, 2] = 0.45
    m["j2dc"][2, :, :, 2] = 0.95
    m["j2dc"][3, 5:, :, 2] = 0.3
    m["j2dc"][3, 60:, :, 2] = 0.9                                # ... and back on camera: a transition step, later the high regime
    a, b = make_net(synth_assets, B), make_net(synth_assets, B)
    a.gravityc = b.gravityc = t(m["gravityc"])
    worst = 0.0
    for i in range(T):
        args = (t(m["j2dc"][:, i]), t(m["accc"][:, i]), t(m["oric"][:, i]))
        pa, ta = a.forward_batch(*args, first_frame=(i == 0))
        pb, tb = b.forward_live(*args, first_frame=(i == 0))
        assert a.get_trace().tolist() == b.get_trace().tolist(), i
        worst = max(worst, maxdiff(pa, pb), maxdiff(ta, tb))
    assert worst <= 1e-5, worst
    assert b.live_replayed() >= 2, b.live_replayed()
    taken, lost = b.live_spin_stats()
    if taken + lost > 0:                                                  # (no AQL chain or no host-writable device memory: nothing is queued ahead)
        assert aql == "1" and taken >= T // 2, (taken, lost)
    for n in ("rnn4", "rnn6", "rnn7"):
        (ha, ca), (hb, cb) = a.get_state(n), b.get_state(n)
        assert maxdiff(ha, hb) <= 1e-5 and maxdiff(ca, cb) <= 1e-5, n


def test_lean_capture_follows_reset_and_parameter_pokes(synth_assets):
    """reset_states() between sequences (init_net must run again: full capture), an attribute poke (re-capture) and a weight
    reload under a live session: the session stays equal to the frame-stepped plan to rounding."""
    T = 50
    m = _mixed_motion(synth_assets, 97, T)
    a, b = make_net(synth_assets, 1), make_net(synth_assets, 1)
    a.gravityc = b.gravityc = t(m["gravityc"])
    b.use_graph = True
    def run(k0, k1, first):
        w = 0.0
        for i in range(k0, k1):
            args = (t(m["j2dc"][0, i]), t(m["accc"][0, i]), t(m["oric"][0, i]))
            pa, ta = a.forward_online(*args, first_frame=(first and i == k0))
            pb, tb = b.forward_online(*args, first_frame=(first and i == k0))
            assert a.get_trace()[0].tolist() == b.get_trace()[0].tolist(), i
            w = max(w, maxdiff(pa, pb), maxdiff(ta, tb))
        return w
    assert run(0, 25, True) <= 1e-5
    a.reset_states(); b.reset_states()
    assert run(3, 30, True) <= 1e-5
    a.use_flat_floor = b.use_flat_floor = False
    assert run(30, T, False) <= 1e-5
    a.load_state_dict(synth_assets["state_dict"]); b.load_state_dict(synth_assets["state_dict"])
    assert run(0, 20, False) <= 1e-5
    assert b.live_stats()[0] > 40


@pytest.mark.parametrize("path", SEQS, ids=[os.path.basename(p)[4:-4] for p in SEQS])
def test_prestep_frames_equal_plain_lean_frames(path, synth_assets, monkeypatch):
    """The idle-time pre-step (rc_live_pre: the recurrent halves of the NEXT frame's layer steps, computed behind a frame when the caller
    paces its frames) forced behind EVERY frame (RC_LIVE_PRESTEP_IDLE_US=0): outputs, branch traces and final states bitwise those of the
    same frames without it -- the K order of every accumulator is kept -- on all reference sequences (regime changes, occlusions,
    init_net, resets of the translation: frames off the lean plan run from the full captures and discard the pre-step)."""
    s = np.load(path)
    live = str(s["live"])
    outs = []
    # (the second leg also leaves the queue armed behind every frame -- rc_aql_arm, a barrier packet the next push releases; the first never)
    # ... and queues the NEXT frame ahead behind every lean frame, its first kernel waiting on the device for the inputs (RC_LIVE_SPIN)
    for env in ({"RC_LIVE_PRESTEP": "0", "RC_LIVE_ARM": "0", "RC_LIVE_SPIN": "0", "RC_LIVE_SPIN_B2B": "0"}, {"RC_LIVE_PRESTEP_IDLE_US": "0", "RC_LIVE_SPIN": "1"}):
        for k in ("RC_LIVE_PRESTEP", "RC_LIVE_PRESTEP_IDLE_US", "RC_LIVE_ARM", "RC_LIVE_SPIN", "RC_LIVE_SPIN_B2B"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        net = make_net(synth_assets, 1, live_ctor=(live == "pre"))
        from robustcap_amd.net.sig_mp import Net
        Net.live = False
        if live == "post":
            net.live = True
        net.use_flat_floor = bool(s["use_flat_floor"])
        net.use_reproj_opt = bool(s["use_reproj_opt"]) if "use_reproj_opt" in s else False
        net.use_vision_updater = bool(s["use_vision_updater"]) if "use_vision_updater" in s else True
        net.use_imu_updater = bool(s["use_imu_updater"]) if "use_imu_updater" in s else True
        net.gravityc = t(s["gravityc"])
        net.use_graph = True
        ft = t(s["first_tran"]) if s["first_tran"].size else None
        T = s["pose"].shape[0]
        poses, trans, traces = [], [], []
        for i in range(T):
            p, tr = net.forward_online(t(s["j2dc"][i]), t(s["accc"][i]), t(s["oric"][i]), ft if i == 0 else None,
                                       bool(s["first_frame"]) and i == 0)
            poses.append(p.clone()), trans.append(tr.clone())
            if i % 7 == 0:
                traces.append(net.get_trace()[0].tolist())
        n_pre, avail = net.live_prestep_stats()
        states = {n: net.get_state(n) for n in ("rnn2", "rnn3", "rnn4", "rnn6", "rnn7", "rnn8")}
        outs.append((torch.stack(poses), torch.stack(trans), traces, states, n_pre, avail, net.live_stats(), net.live_spin_stats()))
        del net
    a, b = outs
    assert a[4] == 0 and a[7] == (0, 0)
    if b[5]:                                                              # (a profiler on the queue, RC_LIVE_AQL=0: no AQL chain, no pre-step)
        assert b[4] >= b[6][0] - 1 and b[4] > 0                           # behind every frame
        assert b[7] == (0, 0) or b[7][0] >= b[6][0] // 2, (b[7], b[6])    # most lean frames started from a kernel that was already waiting
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2] and a[6] == b[6]
    for n in a[3]:
        assert torch.equal(a[3][n][0], b[3][n][0]) and torch.equal(a[3][n][1], b[3][n][1]), n


def test_prestep_is_discarded_by_whatever_touches_the_state(synth_assets, monkeypatch):
    """A live session with a pre-step behind every frame (RC_LIVE_PRESTEP_IDLE_US=0), interleaved with everything that moves the state
    off the frame the pre-step read -- reset_states, an eager batch step, get_state (runs a pending updater step), an attribute poke (ends
    and re-begins the live session), a partial reload of the weights -- against the same script with the pre-step off: bitwise."""
    from robustcap_amd import synth
    from robustcap_amd.net.sig_mp import Net
    body, sd = synth_assets["body"], synth_assets["state_dict"]
    m = synth.make_motion(31, 1, 260, body, conf="mixed")
    script = {40: "reset", 77: "eager", 101: "state", 130: "poke", 171: "reload", 200: "reset", 201: "eager", 202: "state"}
    outs = []
    # third leg: the armed queue alone (a barrier packet behind every frame, no pre-step), through the same script
    # fourth: the frame queued ahead alone (no pre-step): reset, eager steps, pokes and reloads send the waiting kernel away
    for env in ({"RC_LIVE_PRESTEP": "0", "RC_LIVE_ARM": "0", "RC_LIVE_SPIN": "0", "RC_LIVE_SPIN_B2B": "0"}, {"RC_LIVE_PRESTEP_IDLE_US": "0", "RC_LIVE_SPIN": "1"},
                {"RC_LIVE_PRESTEP": "0", "RC_LIVE_PRESTEP_IDLE_US": "0", "RC_LIVE_SPIN": "0", "RC_LIVE_SPIN_B2B": "0"}, {"RC_LIVE_PRESTEP": "0", "RC_LIVE_PRESTEP_IDLE_US": "0", "RC_LIVE_SPIN": "1"}):
        for k in ("RC_LIVE_PRESTEP", "RC_LIVE_PRESTEP_IDLE_US", "RC_LIVE_ARM", "RC_LIVE_SPIN", "RC_LIVE_SPIN_B2B"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        net = Net(body=body, batch=1)
        net.load_state_dict(sd)
        net.gravityc = t(m["gravityc"])
        net.use_graph = True
        res = []
        for i in range(260):
            op = script.get(i)
            if op == "reset":
                net.reset_states()
            elif op == "eager":                                          # a frame through rc_step on the caller's stream
                net.use_graph = False
                p, tr = net.forward_online(t(m["j2dc"][0, i]), t(m["accc"][0, i]), t(m["oric"][0, i]))
                net.use_graph = True
                res.append((p.clone(), tr.clone()))
                continue
            elif op == "state":
                res.append(tuple(x.clone() for x in net.get_state("rnn6")))
            elif op == "poke":
                net.use_flat_floor = False                               # (ends the live session: the frame's arguments are baked in)
            elif op == "reload":
                net.load_state_dict({k: v for k, v in sd.items() if k.startswith("rnn3.")}, strict=False)
            p, tr = net.forward_online(t(m["j2dc"][0, i]), t(m["accc"][0, i]), t(m["oric"][0, i]), None, i == 0 or op == "reset")
            res.append((p.clone(), tr.clone()))
        n_pre, avail = net.live_prestep_stats()
        outs.append((res, n_pre, avail, net.live_stats(), net.live_spin_stats()))
        del net
    a, b, c, d = outs
    assert a[1] == 0 and c[1] == 0 and d[1] == 0 and (not b[2] or b[1] > 150)
    assert a[4] == (0, 0) and c[4] == (0, 0)
    if b[2]:
        assert d[4] == (0, 0) or (d[4][0] > 150 and d[4][1] >= 3), d[4]                      # taken / sent away (the script's resets, eager steps, pokes, reloads)
    for o in (b, c, d):
        assert a[3] == o[3] and len(a[0]) == len(o[0])
        for x, y in zip(a[0], o[0]):
            assert all(torch.equal(u, v) for u, v in zip(x, y))
