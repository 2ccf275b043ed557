#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_resnet_gpu.py -q -x -k "graphed" > $OUT/t_resnet_graph.log 2>&1; tail -3 $OUT/t_resnet_graph.log
timeout 1500 python -m pytest tests/test_dist_gpu.py -q -x -k "graph_replay or two_ranks_on_one_gpu" > $OUT/t_dist.log 2>&1; tail -5 $OUT/t_dist.log
