#!/bin/bash
# Round 4, VERDICT r3 item 1 (re-tile wino4_conv_kernel to 128 co x 16 tiles): what the staging stream of such a tile
# costs on this kernel's schedule, measured.  Timing-only builds (csrc/conv_wino4.hip, W4_EXP_RETILE / W4_EXP_NOXF):
#   rt2 / rt4      weight-slab DMAs x2 / x4 (64 / 128 output channels per workgroup), patch DMAs / 2 / 4, transform kept
#   rt2nx / rt4nx  the same without any input transform (upper bound of sharing V between the waves)
#   nx             the product tile without the input transform (what the transform costs today)
# Usage (GPU box): tools/exp_wino4_retile.sh   ->  table on stdout
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/asvspoof2021_air_amd/_lib
for V in "" nx rt2 rt2nx rt4 rt4nx; do
  if [ -z "$V" ]; then unset AIR_HIP_LIB; echo "== product"; else export AIR_HIP_LIB=$L/libair_hip.$V.so; echo "== $V"; fi
  python tools/kbench_wino.py 64 10 l1,l2,l3,l4 fd 2>&1 | grep -v libdrm
done
