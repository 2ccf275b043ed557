#!/bin/bash
# A/B of the 256 x 256 bf16 GEMM k-loop on one box: product against asvspoof2021_air_amd/_lib/libair_hip.<tag>.so
# Usage (GPU box): bash tools/exp_g2_ab.sh <tag>
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/asvspoof2021_air_amd/_lib
TAG=${1:-g2old}
for V in "" $TAG "" $TAG; do
  if [ -z "$V" ]; then unset AIR_HIP_LIB; echo "== product"; else export AIR_HIP_LIB=$L/libair_hip.$V.so; echo "== $V"; fi
  python tools/kbench_h_gemm.py 2>&1 | grep -v libdrm
done
for V in "" $TAG "" $TAG; do
  if [ -z "$V" ]; then unset AIR_HIP_LIB; echo "== product"; else export AIR_HIP_LIB=$L/libair_hip.$V.so; echo "== $V"; fi
  python bench.py --model ecapa --no-cpu-baseline --no-pmc --no-roofline --no-extra-configs --steps 50 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('timing',{}).get('windows_ms_per_step'))"
done
