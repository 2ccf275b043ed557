#!/bin/bash
# How much does the side-stream overlap of the weight gradients buy (eager), against the one-chain replay?
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "graph AIR_GRAPH=1" "eager_overlap AIR_GRAPH=0 AIR_OVERLAP_WGRAD=1" "eager_serial AIR_GRAPH=0 AIR_OVERLAP_WGRAD=0"; do
  set -- $cfg; name=$1; shift
  env "$@" python bench.py --steps 20 --no-pmc --no-roofline --no-cpu-baseline --no-extra-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$name rep $rep:', d['value'], 'utt/s', d['ms_per_step'], 'ms | host issue', d['host_issue_ms_per_step'], '|', d['launch'][:30])"
done
done
