#!/bin/bash
# LFCC tilings (waves per workgroup x 4-frame groups per wave): us per launch + the LFCC parity tests for each build
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
for rep in 1 2; do
for tag in base n4g2 n4g4 n8g2 n4g1 n2g4; do
  lib=$GRAFT_REPO_ROOT/asvspoof2021_air_amd/_lib/libair_hip.$tag.so
  [ $tag = base ] && lib=$GRAFT_REPO_ROOT/asvspoof2021_air_amd/_lib/libair_hip.so
  echo "== $tag (rep $rep)"; AIR_HIP_LIB=$lib timeout 120 python tools/kbench_lfcc.py 2>&1 | grep -v amdgpu
done
done | tee $OUT/lfcc_variants.log
for tag in n4g2 n4g4; do
  AIR_HIP_LIB=$GRAFT_REPO_ROOT/asvspoof2021_air_amd/_lib/libair_hip.$tag.so timeout 600 python -m pytest tests/test_lfcc_gpu.py -q -x 2>&1 | tail -2
done
