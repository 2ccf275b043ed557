#!/bin/bash
# Copy the judged summaries of a tools/profile_round.sh run from gpurun_out/<tag>/ (scratch) into profiles/ (tracked).
# Usage: tools/collect_profiles.sh <tag>
T=$1; S=gpurun_out/$T
mkdir -p profiles/${T}_bench
cp $S/bench_*.json profiles/${T}_bench/
for k in resnet ecapa resnet_serial; do cp $S/${k}_kernel_stats.md profiles/${T}_${k}_kernel_stats.md; done
python tools/pmc_traffic.py $T > /dev/null
ls profiles | grep "^${T}_"
