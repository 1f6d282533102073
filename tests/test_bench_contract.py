"""bench.py's contract with the driver: the JSON line must come out for the driver's own arguments
(`--gpus 1 --steps 20 --warmup 5` killed round 1's run inside cpu_baseline), whatever the side legs do."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_baseline_on_short_inputs(synth_assets):
    """cpu_baseline must work on inputs shorter than its default sample (T = 5 << 1 + 96 frames)."""
    import bench
    m = bench.make_inputs(synth_assets["body"], 2, 5, "mixed", seed=2)
    out = bench.cpu_baseline(synth_assets["state_dict"], synth_assets["body"], m)
    assert out["kind"] == "port" and out["value"] > 0 and out["cores"] >= 1
    assert "x 4 frames" in out["sample"] and "(4 frames" in out["sample"]          # clamped to T - 1
    assert out["samples"] == 5 and out["min"] <= out["value"] <= out["max"]        # median of five samples
    with pytest.raises(ValueError):
        bench.cpu_baseline(synth_assets["state_dict"], synth_assets["body"], {k: v[:, :1] if v.ndim > 2 else v for k, v in m.items()})


def test_guarded_records_the_error_instead_of_raising():
    import bench

    def boom():
        raise IndexError("index 25 is out of bounds")
    out = bench.guarded(boom)
    assert out == {"error": "IndexError: index 25 is out of bounds"}
    assert bench.guarded(lambda: 7) == 7


def test_pmc_traffic_is_keyed_by_batch_and_schedule():
    import bench
    v, src = bench.pmc_traffic(12345, "mixed", 128)            # no PMC pass of such a batch is committed
    assert v is None and src is None
    for name in os.listdir(os.path.join(ROOT, "profiles")):
        if "pmc_traffic" in name and name.endswith(".json"):
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            if "batch" in d and "conf" in d and "steps" in d:      # (records of earlier rounds carry no frames-per-call key: never quoted)
                v, src = bench.pmc_traffic(int(d["batch"]), d["conf"], int(d["steps"]))
                assert v is not None and v > 0
                assert bench.pmc_traffic(int(d["batch"]), d["conf"], int(d["steps"]) + 1) == (None, None)


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--scaling", "strong"]])
def test_bench_runs_with_the_drivers_arguments(extra):
    """`python bench.py --gpus 1 --steps 20 --warmup 5` -> rc 0 and one JSON line with roofline + cpu_baseline."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"] + extra
    if extra:
        cmd.append("--no-cpu-baseline")                         # the CPU leg is covered by the first run
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["unit"] == "body-frames/s" and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["value"] > 0 and abs(d["value"] - d["config"]["bodies_total"] * 1e3 / d["ms_per_step"]) < 0.01 * d["value"]
    assert d["config"]["batch_per_gpu"] == 256 and "batch 256" in d["config"]["workload"]
    assert d["scaling"] == ("strong" if extra else "weak")
    roof = d["roofline"]
    # frac = the kernel's algorithmic FLOPs per launch / its average launch duration, against the roof of the instructions it
    # issues (split products: the dense bf16 MFMA peak / 6): a hard bound -- a kernel above it is not doing the work
    assert "error" not in roof and 0 < roof["frac"] < 1 and roof["bound"] == "mfma"
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 2e-3 and roof["frac_per_launch"] == roof["frac"]
    assert abs(roof["achieved"] - roof["flop_per_launch"] / (roof["avg_launch_us"] * 1e-6) / 1e12) < 0.01 * roof["achieved"]
    assert roof["traffic"] is None or roof["traffic"] > 0
    assert roof["peak"] > roof["peak_fp32_input"] and roof["frac_fp32_roof"] > roof["frac"]      # split products: the bf16 roof / 6
    # launches on two streams overlap: the kernel's busy time is at most the sum of the launch durations, at least half of it
    assert 1.0 <= roof["concurrency"] <= 3.01 and roof["frac"] <= roof["union"]["frac"] + 1e-4 and roof["union"]["frac"] < 1
    assert abs(roof["avg_launch_us"] * roof["launches"] * 1e-3 - roof["busy_ms"] * roof["concurrency"]) < 0.02 * roof["busy_ms"]
    tm = d["timing"]
    assert tm["reps"] >= 5 and tm["min_call_ms"] <= tm["call_ms"] <= tm["max_call_ms"]
    assert abs(tm["call_ms"] - d["ms_per_step"] * d["steps"]) < 0.01 * tm["call_ms"]        # value = the median repetition
    var = d["variants"]
    for k in ("high", "fp32_mfma", "mixed_long", "high_long", "occ1024"):
        assert "error" not in var[k] and var[k]["value"] > 0, (k, var[k])
    assert var["mixed_long"]["frames"] == 512 and var["high_long"]["frames"] == 512      # BASELINE configs[1]: 512 frames
    lv = var["live_b1"]
    assert "error" not in lv and 0 < lv["p50_us"] <= lv["p99_us"]
    assert lv["lean_frames"] > 0.9 * lv["frames"] and lv["launches_per_lean_frame"] == 7 and lv["dispatch"]
    pc = lv["paced_60fps"]                                            # config 5 as stated: a frame every 16.67 ms, latency from its arrival
    assert pc["frames"] >= 900 and 0 < pc["p50_us"] <= pc["p99_us"] and abs(pc["period_ms"] - 16.667) < 0.01
    if "graph_replay" in lv:
        assert 0 < lv["graph_replay"]["p50_us"] <= lv["graph_replay"]["p99_us"]
    if not extra:
        cpu = d["cpu_baseline"]
        assert "error" not in cpu and cpu["value"] > 0 and cpu["kind"] == "port"
