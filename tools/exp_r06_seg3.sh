#!/bin/bash
# Cost of cutting the captured step into segments, world = 1 (no collective): ms per step and host issue time
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
for rep in 1 2; do
for seg in 0 1; do
  AIR_GRAPH_SEGMENTS=$seg timeout 300 python bench.py --steps 20 --no-pmc --no-roofline --no-cpu-baseline --no-extra-configs > $OUT/n1_seg${seg}_$rep.json 2> $OUT/n1_seg${seg}_$rep.err
  python - <<P
import json
d=json.load(open("$OUT/n1_seg${seg}_$rep.json"))
print("segments=$seg rep $rep:", d["launch"][:40], "| utt/s", d["value"], "| ms/step", d["ms_per_step"], "| host issue ms", d["host_issue_ms_per_step"])
P
done
done
AIR_GRAPH_SEGMENTS=1 timeout 300 python bench.py --model ecapa --steps 20 --no-pmc --no-roofline --no-cpu-baseline --no-extra-configs > $OUT/n1_ecapa_seg1.json 2> $OUT/n1_ecapa_seg1.err
python - <<P
import json
d=json.load(open("$OUT/n1_ecapa_seg1.json"))
print("ecapa segments=1:", d["launch"][:40], "| utt/s", d["value"], "| ms/step", d["ms_per_step"], "| host issue ms", d["host_issue_ms_per_step"])
P
