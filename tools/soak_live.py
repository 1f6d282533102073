#!/usr/bin/env python3
"""Soak of the two round-4 host mechanisms: (1) the lean live frame as an AQL packet chain with completion by the word the last kernel stores --
N frames back to back through the C ABI, outputs finite, the same inputs give the same outputs on a second context (bitwise), device memory
unchanged; (2) rc_smplify_run_batch with the rows as fibers -- the same batch R times: outputs bitwise equal every time, arenas not regrown.
    python tools/soak_live.py [frames=100000] [batch_repeats=30]"""
import ctypes as C
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from robustcap_amd import synth  # noqa: E402
from robustcap_amd.net.sig_mp import Net  # noqa: E402
from robustcap_amd.smplify import TemporalSMPLify  # noqa: E402
import smplify_bench as sb  # noqa: E402


def live_run(sd, body, m, n):
    t = torch.from_numpy
    net = Net(body=body, batch=1)
    net.load_state_dict(sd)
    net.gravityc = t(m["gravityc"])
    net.use_graph = True
    T = m["j2dc"].shape[1]
    ins = [(t(m["j2dc"][0, k]).contiguous(), t(m["accc"][0, k]).contiguous(), t(m["oric"][0, k]).contiguous()) for k in range(T)]
    pose, tran = torch.empty(1, 24, 3, 3), torch.empty(1, 3)
    net.forward_online(*ins[0], first_frame=True)
    fn, ctx = net._lib.rc_live_step, net._ctx
    pp, pt = C.c_void_p(pose.data_ptr()), C.c_void_p(tran.data_ptr())
    ptrs = [(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(c.data_ptr())) for a, b, c in ins]
    h = hashlib.sha256()
    for i in range(n):
        a, b, c = ptrs[1 + i % (T - 1)]
        rc = fn(ctx, a, b, c, None, 0, pp, pt)
        assert rc == 0, (i, rc)
        if i % 97 == 0:
            assert bool(torch.isfinite(pose).all()) and bool(torch.isfinite(tran).all()), i
            h.update(pose.numpy().tobytes()); h.update(tran.numpy().tobytes())
    lean, full = net.live_stats()
    hs = hashlib.sha256(net.get_state("rnn4")[0].cpu().numpy().tobytes()).hexdigest()[:16]
    return {"outputs_sha": h.hexdigest()[:16], "rnn4_h_sha": hs, "lean_frames": lean, "full_frames": full}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    sd, body = synth.make_state_dict(0), synth.make_body(1)
    m = synth.make_motion(7, 1, 600, body, conf="mixed")
    out = {}
    free0 = torch.cuda.mem_get_info()[0]
    a = live_run(sd, body, m, n)
    b = live_run(sd, body, m, n)
    out["live"] = {"frames": n, "first": a, "second_context_equal": a == b}
    assert a == b, (a, b)
    if reps <= 0:
        print(json.dumps(out))
        return
    runner = TemporalSMPLify(body=body, gmm=synth.make_gmm(3))
    rows = [sb.make_case(runner, body, T, seed=seed) for seed, T in ((11, 200), (23, 300), (31, 137), (47, 264), (53, 80), (59, 600))]
    shas, free = set(), []
    for r in range(reps):
        res = runner.run_batch(rows, lr=0.001)
        h = hashlib.sha256()
        for p, tr, upd in res:
            h.update(p.cpu().numpy().tobytes()); h.update(tr.cpu().numpy().tobytes()); h.update(upd.numpy().tobytes())
        shas.add(h.hexdigest()[:16])
        free.append(torch.cuda.mem_get_info()[0])
    out["smplify_batch"] = {"repeats": reps, "distinct_outputs": len(shas), "sha": sorted(shas)[0], "rounds": runner.last_batch_info[0]["rounds"],
                            "free_after_first_MB": round(free[0] / 2**20, 1), "free_after_last_MB": round(free[-1] / 2**20, 1)}
    assert len(shas) == 1 and free[-1] >= free[0] - (8 << 20)
    del runner
    out["device_free_MB"] = {"start": round(free0 / 2**20, 1), "end": round(torch.cuda.mem_get_info()[0] / 2**20, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
