"""rc_live_begin compares the AQL packet chain with the graph replay of the same lean frame before it trusts the chain (rc_api.cpp:
live_selfcheck; round-4/5 review). The loop it protects is live_server.py:40-48 (one frame per call, forward_online -> pose).

Three processes (the switch is read once per process): RC_LIVE_AQL_SELFCHECK=0 (no check), default (check passes: chain in use), =2 (forced
mismatch: the chain is dropped, frames replay the graph). All three must produce the same bits on the same frames -- the check leaves no
trace in the state it ran on, and the fallback is the same arithmetic."""
import hashlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, hashlib, ctypes as C, torch
sys.path.insert(0, %r)
from robustcap_amd import synth
from robustcap_amd.net.sig_mp import Net
sd, body = synth.make_state_dict(0), synth.make_body(1)
T = 48
m = synth.make_motion(11, 1, T, body, conf="mixed")
t = torch.from_numpy
net = Net(body=body, batch=1)
net.load_state_dict(sd)
net.gravityc = t(m["gravityc"])
h = hashlib.sha256()
for i in range(T):
    p, tr = net.forward_live(t(m["j2dc"][:, i]), t(m["accc"][:, i]), t(m["oric"][:, i]), t(m["first_tran"]) if i == 0 else None, i == 0)
    h.update(p.cpu().numpy().tobytes()); h.update(tr.cpu().numpy().tobytes())
cap, aql, note = C.c_int32(0), C.c_int32(0), C.create_string_buffer(256)
net._lib.rc_get_live_backend(net._ctx, C.byref(cap), C.byref(aql), note, 256)
lean, full = net.live_stats()
print("RESULT", h.hexdigest(), cap.value, aql.value, lean, full, "|", note.value.decode())
"""


def _run(mode):
    env = dict(os.environ)
    env.pop("RC_LIVE_AQL_SELFCHECK", None)
    if mode is not None:
        env["RC_LIVE_AQL_SELFCHECK"] = str(mode)
    r = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1]
    head, note = line.split("|", 1)
    _, digest, cap, aql, lean, full = head.split()
    return digest, int(cap), int(aql), int(lean), int(full), note.strip()


def test_selfcheck_passes_forced_mismatch_falls_back_and_neither_changes_a_bit():
    plain = _run(0)
    checked = _run(None)
    forced = _run(2)
    assert plain[1] == 1 and checked[1] == 1 and forced[1] == 1                 # the lean frame is captured in all three
    if plain[2] == 0:
        pytest.skip(f"no AQL chain on this box ({plain[5]}): nothing to self-check")
    assert checked[2] == 1 and checked[5] == "", checked                        # the check passed: chain in use, no note
    assert forced[2] == 0 and "self-check" in forced[5], forced                 # the mismatch path: chain dropped, the note says why
    assert plain[3] > 0 and checked[3] == plain[3] and forced[3] == plain[3]    # the same frames took the lean plan
    assert checked[0] == plain[0], "the self-check left a trace in the state"
    assert forced[0] == plain[0], "graph replay after the fallback differs from the packet chain"
