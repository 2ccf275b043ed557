#!/bin/bash
# Round 4: why was the hipGraph replay of the ECAPA step slower than eager?  Four arms at feat_len 750 and 401:
# eager / graph x weight gradients on the side stream / on the main stream.  Usage (GPU box): tools/exp_ecapa_graph.sh
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
for FL in 750 401; do
  for G in 0 1; do
    for OV in 1 0; do
      echo "== feat_len $FL AIR_GRAPH=$G AIR_OVERLAP_WGRAD=$OV"
      AIR_GRAPH=$G AIR_OVERLAP_WGRAD=$OV python bench.py --model ecapa --feat-len $FL --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-pmc --no-extra-configs 2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print(d['value'], 'utt/s', d['ms_per_step'], 'ms/step host', d['host_issue_ms_per_step'])
"
    done
  done
done
