#!/usr/bin/env python3
"""Parity margins of the HIP path on the captured reference sequences: per fixture max |tran - ref| (m), max joint-position
error (m), max joint-angle error (deg) and max |h - ref| of rnn4's final state, in every arithmetic / engine the library has:
  fp32_b1      batch 1, fp32-MFMA products, forward_sequence (the wavefront engine for every non-live fixture)
  split_b1     batch 1, split-bf16 products (rc_set_gemm_mode 1), forward_sequence
  split_b64    the fixture as row 17 of a batch-64 context (split products by default from 48 rows; the tick's two wide launches on
               two streams, 64-row tiles), the other 63 rows synthetic motions of their own -- the headline arithmetic and engine
  split_b256   the same as row 17 of a batch-256 context: the shared-weight kernel rc_gemm_lds_kernel, three layer-step streams (round 6)
  live_lean    batch 1, forward_online with use_graph: rc_live_step, steady-state frames on the lean seven-launch capture
Used to A/B numerics-affecting kernel changes against the 1e-4 m / 0.1 deg budget before they are adopted.
    python tools/parity_margins.py > profiles/rNN_parity_margins.json"""
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import sig_mp_oracle as O  # noqa: E402  (checker)
from robustcap_amd import synth  # noqa: E402
from robustcap_amd.net.sig_mp import Net  # noqa: E402

t = torch.from_numpy
ROW, BIG = 17, 64


def make_net(s, sd, body, batch):
    live = str(s["live"])
    Net.live = (live == "pre")
    net = Net(body=body, batch=batch)
    net.load_state_dict(sd)
    if live == "post":
        net.live = True
    for k in ("use_flat_floor", "use_reproj_opt", "use_vision_updater", "use_imu_updater"):
        setattr(net, k, bool(s[k]))
    return net


def margins(ob, s, p, tr, h4):
    rp, rt = t(s["pose"]), t(s["tran"])
    jd = float((ob.forward_kinematics(p, tr)[1] - ob.forward_kinematics(rp, rt)[1]).abs().max())
    return {"tran_m": float((tr - rt).abs().max()), "joint_m": jd, "angle_deg": float(O.rotation_angle_deg(p, rp).max()),
            "h_rnn4": float((h4 - t(s["h_rnn4"])).abs().max())}


def main():
    sd, body = synth.make_state_dict(0), synth.make_body(1)
    ob = O.OracleBody(body)
    seqs = {}
    for path in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "seq_*.npz"))):
        s = np.load(path)
        T = int(s["pose"].shape[0])
        live = str(s["live"])
        ff = bool(s["first_frame"])
        rec = {"T": T}
        try:
            if live != "pre":       # (live set before construction keeps `live` a class attribute: per-frame surface only, see live_lean)
                for name, split in (("fp32_b1", False), ("split_b1", True)):
                    net = make_net(s, sd, body, 1)
                    net.set_gemm_mode(split)
                    net.gravityc = t(s["gravityc"])
                    ft = t(s["first_tran"]).view(1, 3) if s["first_tran"].size else None
                    p, tr = net.forward_sequence(t(s["j2dc"][None]), t(s["accc"][None]), t(s["oric"][None]), first_tran=ft, first_frame=ff)
                    rec[name] = margins(ob, s, p[0].cpu(), tr[0].cpu(), net.get_state("rnn4")[0][:, 0])
                    rec[name]["engine_frames"] = net.sequence_stats()[0]
                    del net
                # row ROW of a batch of BIG (64-row tiles) and of 256 (round 6: the shared-weight kernel on three streams -- the headline engine)
                for big, key in ((BIG, "split_b64"), (256, "split_b256")):
                    mu = synth.make_motion(1000 + T, min(big, 32), T, body, conf="mixed")
                    rep = (big + 31) // 32
                    m = {k: np.concatenate([v] * rep, 0)[:big].copy() for k, v in mu.items()}
                    for k, src in (("j2dc", "j2dc"), ("accc", "accc"), ("oric", "oric")):
                        m[k][ROW] = s[src]
                    m["gravityc"][ROW] = s["gravityc"]
                    net = make_net(s, sd, body, big)
                    assert net.gemm_mode, "batches from 48 rows default to the split products"
                    net.gravityc = t(m["gravityc"])
                    ft = None
                    if s["first_tran"].size:
                        ft = t(m["first_tran"].copy())
                        ft[ROW] = t(s["first_tran"])
                    p, tr = net.forward_sequence(t(m["j2dc"]), t(m["accc"]), t(m["oric"]), first_tran=ft, first_frame=ff)
                    rec[key] = margins(ob, s, p[ROW].cpu(), tr[ROW].cpu(), net.get_state("rnn4")[0][:, ROW])
                    rec[key]["engine_frames"] = net.sequence_stats()[0]
                    rec[key]["kernel"] = net.gemm_kernel_name()
                    del net
            # the live path, frame by frame
            net = make_net(s, sd, body, 1)
            net.gravityc = t(s["gravityc"])
            net.use_graph = True
            ft = t(s["first_tran"]) if s["first_tran"].size else None
            ps, ts = [], []
            for i in range(T):
                p, tr = net.forward_online(t(s["j2dc"][i]), t(s["accc"][i]), t(s["oric"][i]), ft if i == 0 else None, ff and i == 0)
                ps.append(p.clone()), ts.append(tr.clone())
            rec["live_lean"] = margins(ob, s, torch.stack(ps), torch.stack(ts), net.get_state("rnn4")[0][:, 0])
            rec["live_lean"]["lean_frames"], rec["live_lean"]["full_frames"] = net.live_stats()
            del net
        finally:
            Net.live = False
        seqs[os.path.basename(path)] = rec
    worst = {}
    for rec in seqs.values():
        for mode, v in rec.items():
            if isinstance(v, dict):
                w = worst.setdefault(mode, {"tran_m": 0.0, "joint_m": 0.0, "angle_deg": 0.0, "h_rnn4": 0.0})
                for k in w:
                    w[k] = max(w[k], v[k])
    print(json.dumps({"tool": "tools/parity_margins.py: the 13 captured reference sequences against the reference's own outputs, per "
                              "arithmetic / engine (see the tool's docstring)",
                      "units": "tran_m / joint_m in metres, angle_deg in degrees, h_rnn4 = max |h - reference| of rnn4's final hidden state",
                      "budget": {"tran_m": 1e-4, "joint_m": 1e-4, "angle_deg": 0.1},
                      "worst": worst, "sequences": seqs}, indent=1))


if __name__ == "__main__":
    main()
