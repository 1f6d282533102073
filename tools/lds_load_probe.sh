#!/bin/bash
# How long does ONE work item of the shared-weight kernel take when 1x / 2x / 3x / 4x as many CUs run the same item beside it?
# (RC_DBG_REPLICATE: one layer step per launch through rc_lstm_step, the launch carrying it n times: 64 / 80 / 16 workgroups x n.)
#   bash tools/lds_load_probe.sh      (on the GPU box, from the repo root; rocprofv3 --kernel-trace)
export TMPDIR=/tmp
for n in 1 2 3 4 6; do
  rm -rf /tmp/lp_$n
  (cd /tmp && RC_DBG_REPLICATE=$n rocprofv3 --kernel-trace -d /tmp/lp_$n -o kt -- python $GRAFT_REPO_ROOT/tools/lds_item_probe.py > /tmp/lp_$n.log 2>&1)
  echo "== replicate $n"
  python tools/lds_item_probe.py --read $(find /tmp/lp_$n -name "*.db" | head -1) | grep "lds_kernel"
done
