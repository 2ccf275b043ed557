#!/bin/bash
# Run on the GPU box (gpurun): bench lines, rocprofv3 kernel traces and PMC traffic passes.
# Usage: tools/profile_round.sh <tag>   -> gpurun_out/<tag>/
set -u
TAG=${1:-r05_final}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 300 python bench.py > $OUT/bench_resnet.json 2> $OUT/bench_resnet.err
timeout 300 python bench.py --feat-len 401 --no-cpu-baseline --no-extra-configs --no-pmc > $OUT/bench_resnet_t401.json 2>> $OUT/bench_resnet.err
timeout 300 python bench.py --model ecapa --steps 20 > $OUT/bench_ecapa_bf16.json 2> $OUT/bench_ecapa.err
timeout 300 python bench.py --model ecapa --steps 20 --feat-len 401 --no-pmc > $OUT/bench_ecapa_bf16_t401.json 2>> $OUT/bench_ecapa.err
timeout 300 python bench.py --model ecapa --steps 20 --augment --no-roofline > $OUT/bench_ecapa_bf16_aug.json 2>> $OUT/bench_ecapa.err
timeout 300 python bench.py --model ecapa --dtype bf16c --steps 8 --no-roofline > $OUT/bench_ecapa_bf16c.json 2>> $OUT/bench_ecapa.err
timeout 300 python bench.py --model ecapa --dtype fp32 --steps 8 --no-roofline > $OUT/bench_ecapa_fp32.json 2>> $OUT/bench_ecapa.err
for m in resnet ecapa; do
  rm -rf $OUT/prof_$m
  timeout 400 rocprofv3 --kernel-trace -d $OUT/prof_$m -o $m -- python bench.py --model $m --plain-timing --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-pmc > $OUT/prof_$m.log 2>&1
  DB=$(find $OUT/prof_$m -name "*.db" | head -1)
  python tools/prof_summary.py $DB $OUT/${m}_kernel_stats.md > /dev/null
  find $OUT/prof_$m -name "*.db" -delete
done
# the same command with the side-stream overlap of the weight-gradient kernels off: kernels run one at a time,
# which is what bench.py's roofline leg times (compare avg us here with roofline.avg_launch_ms)
rm -rf $OUT/prof_resnet_serial
AIR_OVERLAP_WGRAD=0 timeout 400 rocprofv3 --kernel-trace -d $OUT/prof_resnet_serial -o resnet -- python bench.py --plain-timing --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-pmc > $OUT/prof_resnet_serial.log 2>&1
DB=$(find $OUT/prof_resnet_serial -name "*.db" | head -1)
python tools/prof_summary.py $DB $OUT/resnet_serial_kernel_stats.md > /dev/null
find $OUT/prof_resnet_serial -name "*.db" -delete
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  for m in resnet ecapa; do
    rm -rf $OUT/pmc_${m}_$c
    timeout 600 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_${m}_$c -o pmc -- python bench.py --model $m --plain-timing --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extra-configs --no-pmc > $OUT/pmc_${m}_$c.log 2>&1
    DB=$(find $OUT/pmc_${m}_$c -name "*.db" | head -1)
    python tools/pmc_query.py $DB > $OUT/pmc_${m}_$c.txt 2>&1
    find $OUT/pmc_${m}_$c -name "*.db" -delete
  done
done
tail -c 600 $OUT/bench_resnet.json; echo; cut -c1-200 $OUT/bench_resnet_t401.json; cut -c1-200 $OUT/bench_ecapa_bf16.json; cut -c1-200 $OUT/bench_ecapa_bf16_t401.json; cut -c1-200 $OUT/bench_ecapa_bf16_aug.json; cut -c1-200 $OUT/bench_ecapa_bf16c.json; cut -c1-200 $OUT/bench_ecapa_fp32.json
grep -h "wino4_conv\|c1b_" $OUT/pmc_*_SIZE.txt | head -20
