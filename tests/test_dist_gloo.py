"""N > 1 path on CPU: world_size 2 over gloo. Sharded execution + final gather == unsharded.

The per-rank step here is the CPU oracle (no GPU in this container); on the GPU box the same
``robustcap_amd.dist`` helpers wrap ``Net.forward_sequence`` (bench.py, tests/test_gpu_scale.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from robustcap_amd import dist as rdist
from robustcap_amd import synth


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]

def _spawn(fn, args, nprocs, budget_s=100):
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



def _run_rows(sd, body, m, a, b, T):
    from oracle import sig_mp_oracle as O
    net = O.OracleNet(body, batch=b - a)
    net.load_numpy_state_dict(sd)
    net.gravityc = torch.from_numpy(m["gravityc"][a:b])
    t = torch.from_numpy
    P, Tr = [], []
    for i in range(T):
        p, tr = net.forward_batch(t(m["j2dc"][a:b, i]), t(m["accc"][a:b, i]), t(m["oric"][a:b, i]), None, i == 0)
        P.append(p), Tr.append(tr)
    return torch.stack(P, 1), torch.stack(Tr, 1)


def _worker(rank, world, port, B, T, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    r, w, _ = rdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    sd, body = synth.make_state_dict(0), synth.make_body(1)
    m = synth.make_motion(44, B, T, body, conf="mixed")
    rows = torch.arange(B)

    def step(idx):
        a, b = int(idx[0]), int(idx[-1]) + 1
        return _run_rows(sd, body, m, a, b, T)

    pose, tran = rdist.run_sharded(step, [rows], B)
    assert pose.shape == (B, T, 24, 3, 3) and tran.shape == (B, T, 3)
    a, b = rdist.shard_range(B, rank, world)                       # gather-to-root variant (bench.py uses it)
    g = rdist.gather_rows(tran[a:b].contiguous(), B, dst=0)
    assert (g is None) == (rank != 0) and (rank != 0 or torch.equal(g, tran))
    # asynchronous equal-block gathers, several in flight (bench.py's chunked output exchange)
    blocks = [torch.full((2, 5), float(10 * rank + c)) for c in range(3)]
    pending = [rdist.RowGather(blk, dst=0) for blk in blocks]
    for c, h in enumerate(pending):
        out = h.result()
        assert (out is None) == (rank != 0)
        if rank == 0:
            assert out.shape == (2 * world, 5) and all(float(out[2 * q, 0]) == 10 * q + c for q in range(world))
    # fewer rows than ranks (evaluate.run_dataset with a short row list): the empty shard still joins the collective,
    # with the explicit trailing width run_dataset passes (reshape(0, -1) would have raised before reaching it)
    a1, b1 = rdist.shard_range(1, rank, world)
    empty_ok = torch.zeros(max(b1 - a1, 1), 7, 3)[:b1 - a1].reshape(b1 - a1, 21) + float(rank + 1)
    g1 = rdist.gather_rows(empty_ok, 1)
    assert g1.shape == (1, 21) and float(g1[0, 0]) == 1.0
    torch.save((pose, tran), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_equals_unsharded_world2(tmp_path):
    B, T, world = 3, 4, 2                                    # uneven shards: 2 + 1 rows
    port = _free_port()
    _spawn(_worker, (world, port, B, T, str(tmp_path)), world)
    sd, body = synth.make_state_dict(0), synth.make_body(1)
    m = synth.make_motion(44, B, T, body, conf="mixed")
    ref_p, ref_t = [], []
    for r in range(world):                                   # same row blocks, one process
        a, b = rdist.shard_range(B, r, world)
        p, tr = _run_rows(sd, body, m, a, b, T)
        ref_p.append(p), ref_t.append(tr)
    ref_p, ref_t = torch.cat(ref_p), torch.cat(ref_t)
    got = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    assert torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1])   # every rank holds the same full result
    # same row blocks computed in this process (oneDNN thread count differs from the workers': oracle noise only)
    assert float((got[0][0] - ref_p).abs().max()) < 2e-5 and float((got[0][1] - ref_t).abs().max()) < 2e-5
    # and the row blocks agree with one unsharded batch (rows are independent) to oracle noise
    full_p, full_t = _run_rows(sd, body, m, 0, B, T)
    assert float((full_p - ref_p).abs().max()) < 2e-5 and float((full_t - ref_t).abs().max()) < 2e-5


def test_gather_rows_is_identity_without_process_group():
    x = torch.arange(12.0).view(4, 3)
    assert torch.equal(rdist.gather_rows(x, 4), x)
    assert rdist.init_from_env() == (0, 1, 0) or int(os.environ.get("WORLD_SIZE", "1")) > 1
