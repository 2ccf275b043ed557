#!/bin/bash
# Round 6, first GPU trip: new tests, grid-barrier micro-benchmark, baseline bench of both models.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 120 tools/ubench/grid_barrier > $OUT/grid_barrier.log 2>&1
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "prepack" -x > $OUT/t_prepack.log 2>&1
timeout 600 python -m pytest tests/test_dataset_gpu.py tests/test_ecapa_gpu.py -q -x -k "forked or cache or long_input" > $OUT/t_new.log 2>&1
timeout 300 python bench.py --model ecapa --steps 20 --no-pmc > $OUT/bench_ecapa.json 2> $OUT/bench_ecapa.err
KSTAT_N=40 tools/kstat.sh r06_a/ecapa_k -- python bench.py --model ecapa --plain-timing --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-pmc --no-roofline
cat $OUT/grid_barrier.log; tail -3 $OUT/t_prepack.log; tail -3 $OUT/t_new.log; cut -c1-300 $OUT/bench_ecapa.json
