#!/usr/bin/env python3
"""Strong split of ONE batch over the ranks of a process group == the 1-rank result, bit for bit.

    RC_DIST_SHARE_DEVICE=1 RC_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
        --master-addr 127.0.0.1 --master-port 29533 tools/strong_split_check.py [bodies] [frames]

Every rank runs its dist.shard_range block of the same `bodies` sequences (bench.py --scaling strong does exactly this),
the blocks are gathered to rank 0 (dist.gather_rows: blocks may differ by one row), and rank 0 compares with the whole
batch run in one context. With RC_DIST_SHARE_DEVICE=1 all ranks use GPU 0 (1-GPU boxes; the collective then goes over
gloo, since RCCL wants one device per rank). Prints one JSON line on rank 0; exit code 1 on a mismatch."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from robustcap_amd import dist as rdist  # noqa: E402
from robustcap_amd import synth  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 37
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    rank, world, local = rdist.init_from_env()
    torch.cuda.set_device(local)
    from robustcap_amd.net.sig_mp import Net
    sd, body = synth.make_state_dict(0), synth.make_body(1)
    m = bench.make_inputs(body, B, T, "mixed", seed=2)               # the SAME bodies on every rank; each takes its block
    t = torch.from_numpy

    def run(a, b):
        net = Net(body=body, batch=b - a)
        net.load_state_dict(sd)
        net.set_gemm_mode(True)                                       # one product arithmetic whatever the block size
        net.gravityc = t(m["gravityc"][a:b])
        p, tr = net.forward_sequence(t(m["j2dc"][a:b]), t(m["accc"][a:b]), t(m["oric"][a:b]), first_tran=t(m["first_tran"][a:b]))
        torch.cuda.synchronize()
        return p, tr

    a, b = rdist.shard_range(B, rank, world)
    p, tr = run(a, b)
    gp = rdist.gather_rows(p.reshape(b - a, -1), B, dst=0)
    gt = rdist.gather_rows(tr.reshape(b - a, -1), B, dst=0)
    ok = True
    if rank == 0:
        fp, ft = run(0, B)
        ok = bool(torch.equal(gp.view_as(fp), fp) and torch.equal(gt.view_as(ft), ft))
        print(json.dumps({"world": world, "bodies": B, "frames": T, "blocks": [rdist.shard_range(B, r, world) for r in range(world)],
                          "bitwise_equal": ok}), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
