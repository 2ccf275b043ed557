#!/bin/bash
# Launch modes with world = 2 on ONE GPU over gloo (transport not representative; host issue time and call pattern are)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT AIR_DIST_BACKEND=gloo
run() {  # name, env...
  name=$1; shift
  env "$@" AIR_BENCH_JSON_OUT=$OUT/n2_$name.json timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 500)) bench.py --gpus 2 --steps 10 --warmup 3 --plain-timing --no-roofline --no-extra-configs > $OUT/n2_$name.out 2> $OUT/n2_$name.err
  python - <<P
import json
d=json.load(open("$OUT/n2_$name.json"))
print("$name", d["launch"][:60], "| ms/step", d["ms_per_step"], "| host issue ms", d["host_issue_ms_per_step"], "| comm", d["ddp"]["communication"], "| buckets", d["ddp"]["buckets_in_backward"], d["ddp"]["buckets_between_replays"])
P
}
run eager AIR_GRAPH=0
run chain AIR_GRAPH=1 AIR_GRAPH_SEGMENTS=0
run segments AIR_GRAPH=1
run eager2 AIR_GRAPH=0
run segments2 AIR_GRAPH=1
