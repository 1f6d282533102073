#!/usr/bin/env python3
"""Parity margins of the HIP path on the captured reference sequences: per fixture max |tran - ref| (m), max joint-position
error (m) and max joint-angle error (deg), for the build in the tree. Used to A/B numerics-affecting kernel changes
(e.g. a faster gate epilogue) against the 1e-4 m / 0.1 deg budget before they are adopted."""
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


def main():
    sd, body = synth.make_state_dict(0), synth.make_body(1)
    ob = O.OracleBody(body)
    out = {}
    for path in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "seq_*.npz"))):
        s = np.load(path)
        live = str(s["live"])
        Net.live = (live == "pre")
        net = Net(body=body, batch=1)
        Net.live = False
        net.load_state_dict(sd)
        net.set_gemm_mode(os.environ.get("PM_SPLIT", "0") == "1")
        if live == "post":
            net.live = True
        for k in ("use_flat_floor", "use_reproj_opt", "use_vision_updater", "use_imu_updater"):
            setattr(net, k, bool(s[k]))
        net.gravityc = t(s["gravityc"])
        ft = t(s["first_tran"]).view(1, 3) if s["first_tran"].size else None
        p, tr = net.forward_sequence(t(s["j2dc"][None]), t(s["accc"][None]), t(s["oric"][None]), first_tran=ft, first_frame=bool(s["first_frame"]))
        p, tr = p[0].cpu(), tr[0].cpu()
        rp, rt = t(s["pose"]), t(s["tran"])
        jd = float((ob.forward_kinematics(p, tr)[1] - ob.forward_kinematics(rp, rt)[1]).abs().max())
        out[os.path.basename(path)] = {"T": int(p.shape[0]), "tran_m": float((tr - rt).abs().max()), "joint_m": jd,
                                       "angle_deg": float(O.rotation_angle_deg(p, rp).max()),
                                       "h_rnn4": float((net.get_state("rnn4")[0][:, 0] - t(s["h_rnn4"])).abs().max())}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
