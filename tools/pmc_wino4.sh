#!/bin/bash
# Run on the GPU box: HBM-side traffic and L2 hit rate of wino4_conv_kernel per LAYER (one kbench_wino.py run per
# layer and pass, so template instances that serve two layers are told apart).
# Usage: tools/pmc_wino4.sh <tag> [env assignments...]        -> gpurun_out/<tag>/pmc_wino4.txt
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
: > $OUT/pmc_wino4.txt
for L in l1 l2 l3 l4; do
  for PASS in f r; do
    for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
      rm -rf $OUT/p
      timeout 200 rocprofv3 --kernel-trace --pmc $P -d $OUT/p -o pmc -- python tools/kbench_wino.py 64 3 $L $PASS > $OUT/p.log 2>&1
      DB=$(find $OUT/p -name "*.db" | head -1)
      echo "== $L $PASS [$*]" >> $OUT/pmc_wino4.txt
      if [ -n "$DB" ]; then python tools/pmc_query.py $DB wino4_conv >> $OUT/pmc_wino4.txt 2>&1; else echo "pass failed: $P" >> $OUT/pmc_wino4.txt; tail -3 $OUT/p.log >> $OUT/pmc_wino4.txt; fi
      rm -rf $OUT/p
    done
  done
done
cat $OUT/pmc_wino4.txt
