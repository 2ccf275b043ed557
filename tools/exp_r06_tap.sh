#!/bin/bash
# Round 6: Res2-chain prologues - parity tests, then A/B of the ECAPA bf16 step (AIR_TAP_PROLOGUE=0/1/2/3, alternating).
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ecapa_bf16_gpu.py -q -x -k "prologue or fused_res2 or row_piece" > $OUT/t_pro.log 2>&1
tail -3 $OUT/t_pro.log
for rep in 1 2; do
  for v in 0 1 2 3; do
    AIR_TAP_PROLOGUE=$v timeout 300 python bench.py --model ecapa --steps 20 --no-pmc --no-roofline --no-cpu-baseline > $OUT/bench_pro${v}_$rep.json 2> $OUT/bench_pro${v}_$rep.err
    echo "AIR_TAP_PROLOGUE=$v rep $rep: $(cut -c83-130 $OUT/bench_pro${v}_$rep.json)"
  done
done
for v in 0 1; do
AIR_TAP_PROLOGUE=0 AIR_TAP_ROWS=$v timeout 300 python bench.py --model ecapa --steps 20 --no-pmc --no-roofline --no-cpu-baseline > $OUT/bench_rows$v.json 2> $OUT/bench_rows$v.err
echo "AIR_TAP_PROLOGUE=0 AIR_TAP_ROWS=$v: $(cut -c83-130 $OUT/bench_rows$v.json)"
done
