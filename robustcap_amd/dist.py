"""Multi-GPU execution of the sig_mp path: shard independent sequences over ranks, gather the results once.

The path shards embarrassingly -- every (sequence, camera) is an independent recurrence and the reference runs
them strictly one after another (evaluate.py:66,75) -- so there is NO data-path collective: each rank owns a
contiguous block of rows, a full copy of the weights (242 MB) and its own context/stream. The only exchange is
the final all-gather of the outputs (876 B per body-frame) over RCCL/xGMI (backend "nccl" on ROCm) or gloo on
CPU-only hosts (tests).

One process per GPU, launched by torch.distributed.run / torchrun; RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR,
MASTER_PORT are read from the environment.
"""
import os

import torch
import torch.distributed as dist


def shard_range(n_rows, rank, world):
    """Contiguous balanced block [start, stop) of rows for ``rank``; the first n_rows % world ranks get one more."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    base, extra = divmod(n_rows, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def init_from_env(backend=None):
    """Initialise torch.distributed from the launcher's environment. Returns (rank, world, local_rank).
    With WORLD_SIZE unset or 1 nothing is initialised."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("RC_DIST_SHARE_DEVICE") == "1":                  # dry runs of the N-rank flow on a 1-GPU box
        local = 0
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on these hosts
        if backend is None:
            backend = os.environ.get("RC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def gather_rows(local, n_rows_total, dst=None):
    """Gather row blocks of unequal size along dim 0 into the full [n_rows_total, ...] tensor.

    dst=None: all-gather, every rank gets the result. dst=r: gather to rank r only (the other ranks return None) --
    over xGMI that is 7 point-to-point transfers into r instead of a ring pass of everybody's block.
    ``local`` is this rank's block as produced by shard_range (same trailing shape everywhere)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    if local.is_cuda and dist.get_backend() == "gloo":                 # gloo gathers host tensors: stage through the host
        out = gather_rows(local.cpu(), n_rows_total, dst)
        return None if out is None else out.to(local.device)
    sizes = [shard_range(n_rows_total, r, world) for r in range(world)]
    cap = max(b - a for a, b in sizes)
    pad = local.new_zeros((cap,) + tuple(local.shape[1:]))
    pad[:local.shape[0]] = local
    pad = pad.contiguous()
    if dst is None:
        out = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(out, pad)
    else:
        out = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
        dist.gather(pad, out, dst=dst)
        if rank != dst:
            return None
    return torch.cat([o[:b - a] for o, (a, b) in zip(out, sizes)], dim=0)


class RowGather:
    """Asynchronous gather of EQUAL-sized row blocks to rank ``dst``: started right after the producing kernels are
    enqueued, it runs on the collective's own stream (RCCL over xGMI: 7 point-to-point transfers into ``dst``) while the
    caller keeps enqueueing compute; ``result()`` makes the current stream wait and returns the [world * n, ...] tensor on
    ``dst`` (None elsewhere). Under gloo (CPU hosts, dry runs) the block is staged through the host."""

    def __init__(self, local, dst=0):
        self.dst, self.device = dst, local.device
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.stage = dist.get_backend() == "gloo" and local.is_cuda
        self.src = local.contiguous().cpu() if self.stage else local.contiguous()       # kept alive until result()
        self.parts = [torch.empty_like(self.src) for _ in range(self.world)] if self.rank == dst else None
        self.work = dist.gather(self.src, self.parts, dst=dst, async_op=True)

    def result(self):
        self.work.wait()
        if self.rank != self.dst:
            return None
        out = torch.cat(self.parts, dim=0)
        return out.to(self.device) if self.stage else out


def run_sharded(step_fn, inputs, n_rows_total):
    """Run ``step_fn(*row_block_of_each_input)`` on this rank's rows and gather every output over all ranks.

    inputs: tensors whose dim 0 is the global row axis (every rank holds, or can build, the full tensors; only
    the local block is touched). step_fn returns a tensor or a tuple of tensors with the local rows in dim 0."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    a, b = shard_range(n_rows_total, rank, world)
    outs = step_fn(*[None if x is None else x[a:b] for x in inputs])
    single = not isinstance(outs, (tuple, list))
    outs = [outs] if single else list(outs)
    outs = [gather_rows(o, n_rows_total) for o in outs]
    return outs[0] if single else tuple(outs)
