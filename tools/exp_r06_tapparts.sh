#!/bin/bash
# c1b_tap_kernel<true,0> parts by rocprofv3 kernel durations (the python launch loop itself is ~9 us per call)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
for t in base x1 x2 x4 x8 x9 x13 x15; do
  lib=$GRAFT_REPO_ROOT/asvspoof2021_air_amd/_lib/libair_hip.tap$t.so; [ $t = base ] && lib=$GRAFT_REPO_ROOT/asvspoof2021_air_amd/_lib/libair_hip.so
  AIR_HIP_LIB=$lib KSTAT_N=8 tools/kstat.sh r06_f/tap_$t -- python tools/kbench_h_tap.py > /dev/null 2>&1
  echo "$t: $(grep c1b_tap gpurun_out/r06_f/tap_$t/stats.md | head -1)"
done
