for i in 1 2 3; do RC_PROBE_EXTRA=4 python tools/live_period_probe.py 500 2>/dev/null | grep -E "p50_us|p99_us|mean_us" | tr -d '\n'; echo; done
python - <<'PY'
import os, sys, json
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
from robustcap_amd import synth
import live_latency as L
sd, body = synth.make_state_dict(0), synth.make_body(1)
m = synth.make_motion(7, 1, 600, body, conf="mixed")
for rep in range(2):
    for name, env in (("b2b early queue on", {}), ("b2b early queue off", {"RC_LIVE_SPIN_B2B": "0"})):
        net = L.make(sd, body, m, env=env)
        print(name, L.stats(L.run_c(net, m, 3000)), net.live_spin_stats())
        del net
PY
