#!/bin/bash
# Run on the GPU box: per-kernel durations of one command (rocprofv3 --kernel-trace), top 12 lines.
# Usage: tools/kstat.sh <tag> -- <command...>
TAG=$1; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT; rm -rf $OUT/prof
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace -d $OUT/prof -o k -- "$@" > $OUT/log.txt 2>&1
DB=$(find $OUT/prof -name "*.db" | head -1)
python tools/prof_summary.py $DB $OUT/stats.md > /dev/null
rm -rf $OUT/prof
head -${KSTAT_N:-12} $OUT/stats.md | cut -c1-120
