"""BASELINE config 3 at its stated partition, on CPU: 8 sequences x 9 cameras = 72 (sequence, camera) rows sharded over
8 ranks (gloo), the harness's own sharding + single gather (robustcap_amd.evaluate.shard_rows, the function run_dataset
runs on) -- with the CPU oracle standing in for the per-rank HIP compute (no GPU in this container; on the GPU box
tests/test_gpu_scale.py and bench.py drive the same helpers with Net.forward_sequence). Every rank must end up holding all
72 rows, each equal to that row run alone (evaluate.py:66,75 runs them strictly one after another)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from robustcap_amd import dist as rdist
from robustcap_amd import synth

N_SEQ, N_CAM, T, WORLD = 8, 9, 3, 8


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]

def _spawn(fn, args, nprocs, budget_s=150):
    """mp.spawn with a deadline (a normal run takes 10-30 s) and two retries on a fresh port: a rendezvous that never completes (port stolen between
    _free_port() and the bind, a starved host) must not hang the CPU suite. args[1] is the port."""
    import time
    for attempt in range(3):
        ctx = mp.spawn(fn, args=args, nprocs=nprocs, join=False)
        t0 = time.time()
        done = False
        while time.time() - t0 < budget_s:
            if ctx.join(timeout=1.0):
                done = True
                break
        if done:
            return
        for pr in ctx.processes:
            if pr.is_alive():
                pr.terminate()
        for pr in ctx.processes:
            pr.join(10)
        args = (args[0], _free_port()) + tuple(args[2:])
    raise RuntimeError("gloo workers did not finish within the deadline (three attempts)")



def _dataset(body):
    ds = synth.make_dataset(31, N_SEQ, T + 1, body, n_cam=N_CAM, conf="mixed")
    for k in ("pose", "tran", "imu_ori", "imu_acc"):                    # ragged: the last sequence is one frame shorter
        ds[k][N_SEQ - 1] = ds[k][N_SEQ - 1][:T]
    ds["joint2d_mp"][N_SEQ - 1] = ds["joint2d_mp"][N_SEQ - 1][:, :T]
    return ds


def _oracle_compute(ds, sd, body, Tmax):
    """compute(rows) of shard_rows with the CPU oracle: camera inputs (oracle/harness_oracle.py), then the batched net."""
    from oracle import harness_oracle as H
    from oracle import sig_mp_oracle as O

    def compute(mine):
        n = len(mine)
        j2d, acc, ori = torch.zeros(n, Tmax, 33, 3), torch.zeros(n, Tmax, 6, 3), torch.eye(3).expand(n, Tmax, 6, 3, 3).clone()
        grav, ft = torch.zeros(n, 3), torch.zeros(n, 3)
        for r, (i, j) in enumerate(mine):
            k, a, o, g = H.camera_inputs(ds["joint2d_mp"][i][j], ds["imu_acc"][i], ds["imu_ori"][i], ds["cam_K"][i][j], ds["cam_T"][i][j])
            L = k.shape[0]
            j2d[r, :L], acc[r, :L], ori[r, :L], grav[r] = k, a, o, g
            ft[r] = H.first_translation(ds["tran"][i], ds["cam_T"][i][j])
        net = O.OracleNet(body, batch=n)
        net.load_numpy_state_dict(sd)
        net.gravityc = grav
        P, Tr = [], []
        for f in range(Tmax):
            p, tr = net.forward_batch(j2d[:, f], acc[:, f], ori[:, f], ft if f == 0 else None, False)
            P.append(p), Tr.append(tr)
        return torch.stack(P, 1), torch.stack(Tr, 1)
    return compute


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    rdist.init_from_env(backend="gloo")
    from robustcap_amd import evaluate as ev
    sd, body = synth.make_state_dict(0), synth.make_body(1)
    ds = _dataset(body)
    rows = ev.rows_of(ds)
    assert len(rows) == N_SEQ * N_CAM
    Tmax = max(len(ds["pose"][i]) for i, _ in rows)
    rows_out, pose, tran = ev.shard_rows(rows, Tmax, _oracle_compute(ds, sd, body, Tmax), device="cpu")
    assert rows_out == rows and pose.shape == (72, Tmax, 24, 3, 3) and tran.shape == (72, Tmax, 3)
    a, b = rdist.shard_range(len(rows), rank, world)
    assert b - a == 9                                                   # one sequence's nine cameras per rank
    torch.save((pose, tran), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_config3_partition_72_rows_over_8_ranks(tmp_path):
    port = _free_port()
    _spawn(_worker, (WORLD, port, str(tmp_path)), WORLD)
    got = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(WORLD)]
    for r in range(1, WORLD):                                           # every rank holds the same full result
        assert torch.equal(got[0][0], got[r][0]) and torch.equal(got[0][1], got[r][1])
    # ... and it is what one process computes for the same rows (rows are independent: batch composition only moves
    # oneDNN's summation order)
    from robustcap_amd import evaluate as ev
    sd, body = synth.make_state_dict(0), synth.make_body(1)
    ds = _dataset(body)
    rows = ev.rows_of(ds)
    Tmax = max(len(ds["pose"][i]) for i, _ in rows)
    torch.set_num_threads(4)
    ref_p, ref_t = _oracle_compute(ds, sd, body, Tmax)(rows)
    assert float((got[0][0] - ref_p).abs().max()) < 5e-5 and float((got[0][1] - ref_t).abs().max()) < 5e-5
    # a row run alone, like evaluate.py's own loop (a rank's block boundary must not matter)
    one_p, one_t = _oracle_compute(ds, sd, body, Tmax)([rows[40]])
    assert float((got[0][0][40] - one_p[0]).abs().max()) < 5e-5 and float((got[0][1][40] - one_t[0]).abs().max()) < 5e-5
    assert np.isfinite(got[0][0].numpy()).all()
