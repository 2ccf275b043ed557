#!/bin/bash
# ECAPA bf16 hipGraph step: weight gradients queued and handed to a side stream in front of each block's Res2 chain
# (AIR_WGRAD_BATCHED=1, 4 forks + 1 join in the captured graph) against one chain (=0), alternating on one box.
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
for rep in 1 2; do
  for V in 1 0; do
    echo "== AIR_WGRAD_BATCHED=$V"
    AIR_WGRAD_BATCHED=$V python bench.py --model ecapa --no-cpu-baseline --no-pmc --no-roofline --no-extra-configs --steps 50 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_issue_ms_per_step'], d.get('timing',{}).get('windows_ms_per_step'))"
  done
done
for V in 1 0; do
  echo "== T = 401, AIR_WGRAD_BATCHED=$V"
  AIR_WGRAD_BATCHED=$V python bench.py --model ecapa --feat-len 401 --no-cpu-baseline --no-pmc --no-roofline --no-extra-configs --steps 50 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_issue_ms_per_step'])"
done
