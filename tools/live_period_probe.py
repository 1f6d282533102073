#!/usr/bin/env python3
"""How the paced live latency depends on the idle time in front of a frame (config 5): p50 / p99 of rc_live_step at several arrival
periods, with the idle-time pre-step and without it.  python tools/live_period_probe.py [frames=500]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from robustcap_amd import synth  # noqa: E402
import live_latency as L  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    sd, body = synth.make_state_dict(0), synth.make_body(1)
    m = synth.make_motion(7, 1, 600, body, conf=os.environ.get("RC_PROBE_CONF", "mixed"))
    rows = []
    modes = (("prestep", {"RC_LIVE_PRESTEP_IDLE_US": "100", "RC_LIVE_ARM": "0"}), ("plain", {"RC_LIVE_PRESTEP": "0", "RC_LIVE_PRESTEP_IDLE_US": "100", "RC_LIVE_ARM": "0"}),
             ("prestep+arm", {"RC_LIVE_PRESTEP_IDLE_US": "100", "RC_LIVE_ARM": "1"}),
             ("plain+arm", {"RC_LIVE_PRESTEP": "0", "RC_LIVE_PRESTEP_IDLE_US": "100", "RC_LIVE_ARM": "1"}))
    if os.environ.get("RC_PROBE_EXTRA"):                                # side questions on the default configuration, paced
        modes = (("prestep+arm", {"RC_LIVE_PRESTEP_IDLE_US": "100"}), ("edge fences at agent scope", {"RC_LIVE_PRESTEP_IDLE_US": "100", "RC_AQL_EDGE_SCOPE": "agent"}),
                 ("completion by the signal", {"RC_LIVE_PRESTEP_IDLE_US": "100", "RC_LIVE_DONE_FLAG": "0"}), ("prestep+arm again", {"RC_LIVE_PRESTEP_IDLE_US": "100"}))
    if os.environ.get("RC_PROBE_EXTRA") == "4":                         # one paced leg of the default configuration (A/B of library builds through RC_LIB_PATH)
        modes = (("default", {}),)
    if os.environ.get("RC_PROBE_EXTRA") == "3":                         # the next frame's first kernel launched ahead and waiting on the device (RC_LIVE_SPIN)
        modes = (("default", {"RC_LIVE_PRESTEP_IDLE_US": "100"}), ("frame queued ahead, first kernel waiting (paced callers)", {"RC_LIVE_PRESTEP_IDLE_US": "100", "RC_LIVE_SPIN": "1"}),
                 ("... behind every lean frame", {"RC_LIVE_PRESTEP_IDLE_US": "100", "RC_LIVE_SPIN": "2"}), ("default again", {"RC_LIVE_PRESTEP_IDLE_US": "100"}))
    if os.environ.get("RC_PROBE_EXTRA") == "2":                         # timing probe (needs profiles/r05_live_prequeue_experiment.diff applied): the next frame's packets already behind the armed barrier (RC_PROBE_CONF=high)
        modes = (("prestep+arm", {"RC_LIVE_PRESTEP_IDLE_US": "100"}), ("prestep+arm+frame queued ahead", {"RC_LIVE_PRESTEP_IDLE_US": "100", "RC_LIVE_PREQUEUE": "1"}),
                 ("plain+arm", {"RC_LIVE_PRESTEP": "0", "RC_LIVE_PRESTEP_IDLE_US": "100"}),
                 ("plain+arm+frame queued ahead", {"RC_LIVE_PRESTEP": "0", "RC_LIVE_PRESTEP_IDLE_US": "100", "RC_LIVE_PREQUEUE": "1"}))
    for period_ms in ((16.667,) if os.environ.get("RC_PROBE_EXTRA") == "4" else (16.667, 0.0) if os.environ.get("RC_PROBE_EXTRA") == "3" else ((16.667, 1.0) if os.environ.get("RC_PROBE_EXTRA") else (16.667, 1.0, 0.3, 0.0))):
        for name, env in modes:
            net = L.make(sd, body, m, env=env)
            st = L.stats(L.run_c(net, m, n, period_ms * 1e-3))
            rows.append({"period_ms": period_ms, "mode": name, **st, "pre": list(net.live_prestep_stats()) if hasattr(net, "live_prestep_stats") else None,
                         "spin": list(net.live_spin_stats()) if hasattr(net, "live_spin_stats") else None})
            del net
    print(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
