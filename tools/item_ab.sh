#!/bin/bash
# per-variant item durations: tools/item_ab.sh name1 name2 ...   (tools/probe_<name>.so)
export TMPDIR=/tmp
for n in "$@"; do
  rm -rf /tmp/ip_$n
  (cd /tmp && RC_LIB_PATH=$GRAFT_REPO_ROOT/tools/probe_$n.so rocprofv3 --kernel-trace -d /tmp/ip_$n -o kt -- python $GRAFT_REPO_ROOT/tools/lds_item_probe.py > /tmp/ip_$n.log 2>&1)
  echo "== $n"
  python tools/lds_item_probe.py --read $(find /tmp/ip_$n -name "*.db" | head -1) | grep "lds_kernel" | grep -v " 144 wg"
done
