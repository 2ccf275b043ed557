#!/bin/bash
# de-phase the persistent Winograd workgroups (option WINO4_DEPHASE = N x 4096 cycles).
export PYTHONPATH=$GRAFT_REPO_ROOT
for d in 0 2 4 6 8 12; do
  echo "WINO4_DEPHASE=$d"
  AIR_WINO4_DEPHASE=$d python tools/kbench_wino.py 64 20 l1,l2,l3,l4 frd 2>&1 | grep -v "^/opt"
done
