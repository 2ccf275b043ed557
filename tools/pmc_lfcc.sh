#!/bin/bash
# Run on the GPU box: where the LFCC kernel's cycles go (issue vs wait), two PMC passes over tools/kbench_lfcc.py.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-lfcc_pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
: > $OUT/pmc.txt
python tools/kbench_lfcc.py > $OUT/kbench.txt 2>&1
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_WAVES"
 "GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"
)
i=0
for P in "${PASSES[@]}"; do
  rm -rf $OUT/p$i
  timeout 300 rocprofv3 --kernel-trace --pmc $P -d $OUT/p$i -o pmc -- python tools/kbench_lfcc.py > $OUT/p$i.log 2>&1
  DB=$(find $OUT/p$i -name "*.db" | head -1)
  if [ -n "$DB" ]; then python tools/pmc_query.py $DB lfcc >> $OUT/pmc.txt 2>&1; else echo "pass $i failed: $P" >> $OUT/pmc.txt; tail -3 $OUT/p$i.log >> $OUT/pmc.txt; fi
  rm -rf $OUT/p$i
  i=$((i+1))
done
cat $OUT/kbench.txt $OUT/pmc.txt
