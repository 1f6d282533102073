"""The rocprofv3 (rocpd sqlite) readers under tools/ on a hand-made trace: three kernels on two queues with known gaps."""
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make_db(path):
    db = sqlite3.connect(path)
    db.execute("create table rocpd_info_kernel_symbol_x (id integer primary key, kernel_name text)")
    db.execute("create table rocpd_kernel_dispatch_x (id integer primary key, kernel_id integer, queue_id integer, stream_id integer, "
               "start integer, end integer, grid_size_x integer, workgroup_size_x integer)")
    db.executemany("insert into rocpd_info_kernel_symbol_x values (?, ?)", [(1, "gemm(GemmLaunch)"), (2, "tail(FrameBuffers)")])
    rows, t = [], 1_000_000
    for k in range(40):                               # 40 x [gemm 50 us | 2 us gap | tail 10 us | 8 us gap]
        rows.append((1, 1, 1, t, t + 50_000, 256 * 256, 256)); t += 52_000
        rows.append((2, 2, 2, t, t + 10_000, 256 * 64, 64)); t += 18_000
    db.executemany("insert into rocpd_kernel_dispatch_x (kernel_id, queue_id, stream_id, start, end, grid_size_x, workgroup_size_x) "
                   "values (?, ?, ?, ?, ?, ?, ?)", rows)
    db.commit(); db.close()


def _run(tool, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), *args], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return r.stdout


def test_stats_timeline_and_gaps(tmp_path):
    db = str(tmp_path / "t_results.db")
    _make_db(db)
    out = _run("rocpd_stats.py", db)
    lines = out.strip().splitlines()
    assert lines[0].startswith("kernel,calls") and lines[1].startswith("gemm,40,2000.00,50.00") and lines[2].startswith("tail,40,400.00,10.00")
    out = _run("rocpd_timeline.py", db, "0.5", "700")
    assert "idle" in out and "gemm" in out and "avg   50.00 us" in out and "avg   10.00 us" in out
    out = _run("rocpd_gaps.py", db, "1", "-5")        # last 1 ms, every gap >= 5 us in time order: the 8 us gaps only
    gaps = [ln for ln in out.splitlines() if ln.strip().startswith("gap")]
    assert gaps and all(" 8.0 us" in g and "tail" in g.split("->")[0] for g in gaps)
