#!/usr/bin/env python3
"""Soak of the per-row-cursor wavefront engine against the frame-stepped launches: random batches, call lengths and
confidence schedules that switch regime every few frames (many riders, transition rows and init_net waits per call), outputs
and final states compared bit for bit. usage: python tools/soak_engine.py [cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from robustcap_amd import synth  # noqa: E402
from robustcap_amd.net.sig_mp import Net  # noqa: E402


def schedule(rng, B, T):
    """per-row confidences: runs of 1..9 frames in a random regime (occluded / mid / visible), some rows hugging the thresholds"""
    c = np.empty((B, T), np.float32)
    for b in range(B):
        i = 0
        while i < T:
            n = int(rng.integers(1, 10))
            r = rng.random()
            v = rng.uniform(0.3, 0.69) if r < 0.35 else (rng.uniform(0.71, 0.79) if r < 0.55 else rng.uniform(0.81, 0.99))
            if rng.random() < 0.05:
                v = float(rng.choice([0.7, 0.8, 0.69999, 0.70001, 0.79999, 0.80001]))
            c[b, i:i + n] = v
            i += n
    return c


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    sd, body = synth.make_state_dict(0), synth.make_body(1)
    t = torch.from_numpy
    bad = 0
    for case in range(cases):
        B = int(rng.choice([3, 17, 48, 64, 72, 88, 100, 130, 256]))
        T = int(rng.integers(12, 72))
        cut = int(rng.integers(9, T - 2)) if T > 24 and rng.random() < 0.5 else T        # the call split in two
        m = synth.make_motion(1000 + case, min(B, 16), T, body, conf="high")
        rep = (B + 15) // 16
        m = {k: np.concatenate([v] * rep, 0)[:B].copy() for k, v in m.items()}
        m["j2dc"][..., 2] = schedule(rng, B, T)[:, :, None]
        outs = []
        for seq in (True, False):
            net = Net(body=body, batch=B)
            net.load_state_dict(sd)
            net.set_sequence_mode(seq, 8, force=True)
            net.gravityc = t(m["gravityc"])
            P, Tr = [], []
            for lo, hi in ((0, cut), (cut, T)):
                if hi > lo:
                    p, tr = net.forward_sequence(t(m["j2dc"][:, lo:hi]), t(m["accc"][:, lo:hi]), t(m["oric"][:, lo:hi]), first_frame=(lo == 0))
                    P.append(p), Tr.append(tr)
            torch.cuda.synchronize()
            st = [net.get_state(n) for n in ("rnn2", "rnn3", "rnn4", "rnn6", "rnn7", "rnn8")]
            outs.append((torch.cat(P, 1), torch.cat(Tr, 1), st, net.get_trace(), net.sequence_stats()))
        (wp, wt, ws, wtr, wstat), (sp, stt, ss, strc, sstat) = outs
        ok = torch.equal(wp, sp) and torch.equal(wt, stt) and torch.equal(wtr, strc)
        ok = ok and all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(ws, ss))
        ok = ok and bool(torch.isfinite(wp).all()) and wstat[0] > 0 and sstat[0] == 0
        bad += 0 if ok else 1
        print(f"case {case}: B={B} T={T} cut={cut} wave/stepped/ticks={wstat} {'ok' if ok else 'MISMATCH'}", flush=True)
    print("soak:", "all equal" if bad == 0 else f"{bad} mismatching cases")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
