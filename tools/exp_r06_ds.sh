#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_dataset_gpu.py -q -x > $OUT/t_ds.log 2>&1; tail -3 $OUT/t_ds.log
timeout 400 python bench.py --steps 20 --no-pmc --no-roofline --no-cpu-baseline --no-extra-configs --from-dataset > $OUT/bench_ds.json 2> $OUT/bench_ds.err
python - <<P
import json
d=json.load(open("$OUT/bench_ds.json"))
print(d["value"], d["ms_per_step"]); print(json.dumps(d["configs"], indent=1)[:1500])
P
tail -3 $OUT/bench_ds.err
