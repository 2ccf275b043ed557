#!/bin/bash
# kbench_h_gemm.py over the product and every asvspoof2021_air_amd/_lib/libair_hip.<tag>.so named, twice, alternating.
# Usage (GPU box): bash tools/exp_g2_multi.sh <tag> [<tag> ...]
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/asvspoof2021_air_amd/_lib
for rep in 1 2; do
  for V in "" "$@"; do
    if [ -z "$V" ]; then unset AIR_HIP_LIB; echo "== product"; else export AIR_HIP_LIB=$L/libair_hip.$V.so; echo "== $V"; fi
    python tools/kbench_h_gemm.py 2>&1 | grep -v libdrm
  done
done
