#!/bin/bash
# Round 4: layer4's HBM-side fetch by the assignment of the cut items' halves to workgroup indices (wino4_conv_kernel).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
for V in product nosplit; do
  unset AIR_HIP_LIB AIR_WINO4_SPLIT
  [ $V = nosplit ] && export AIR_WINO4_SPLIT=0
  rm -rf /tmp/p; timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p -o pmc -- python tools/kbench_wino.py 64 3 l4 f > /tmp/p.log 2>&1
  DB=$(find /tmp/p -name "*.db" | head -1); echo "== $V"; python tools/pmc_query.py $DB wino4_conv
  python tools/kbench_wino.py 64 20 l3,l4 fd 2>&1 | grep "^l"
done
