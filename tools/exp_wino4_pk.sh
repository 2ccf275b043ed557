#!/bin/bash
# wino4_conv_kernel input transform: product (W4_PK, csrc/conv_wino4.hip) against a variant built with
#   python -m asvspoof2021_air_amd.build --variant pk0 conv_wino4.hip -DW4_PK=0   (scalar; pk2: -DW4_PK=2)
# Usage (GPU box): bash tools/exp_wino4_pk.sh
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/asvspoof2021_air_amd/_lib
python tools/kbench_wino.py 64 10 l1 fd > /dev/null 2>&1
for rep in 1 2; do
for V in "" pk0; do
  if [ -z "$V" ]; then unset AIR_HIP_LIB; echo "== product"; else export AIR_HIP_LIB=$L/libair_hip.$V.so; echo "== $V"; fi
  python tools/kbench_wino.py 64 20 l1,l2,l3,l4 fd 2>&1 | grep -v libdrm
done; done
