#!/bin/bash
# Run on the GPU box: PMC passes over one micro-benchmark command, per-kernel averages of every counter.
# Usage: tools/pmc_kernel.sh <tag> <kernel-name filter> -- <command...>      -> gpurun_out/<tag>/pmc.txt
set -u
TAG=$1; FILT=$2; shift 3
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
: > $OUT/pmc.txt
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
 "TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_SRC_FIFO_FULL_sum"
 "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_BUSY_avr"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_GATE_EN1_sum"
 "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum"
 "GRBM_GUI_ACTIVE TCC_CYCLE_sum"
)
i=0
for P in "${PASSES[@]}"; do
  rm -rf $OUT/p$i
  timeout 300 rocprofv3 --kernel-trace --pmc $P -d $OUT/p$i -o pmc -- "$@" > $OUT/p$i.log 2>&1
  DB=$(find $OUT/p$i -name "*.db" | head -1)
  if [ -n "$DB" ]; then python tools/pmc_query.py $DB "$FILT" >> $OUT/pmc.txt 2>&1; else echo "pass $i failed: $P" >> $OUT/pmc.txt; tail -3 $OUT/p$i.log >> $OUT/pmc.txt; fi
  rm -rf $OUT/p$i
  i=$((i+1))
done
cat $OUT/pmc.txt
