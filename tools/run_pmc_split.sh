#!/bin/bash
# rocprofv3 --pmc passes (one counter group per pass) over tools/gemm_split_pmc.py
# usage: tools/run_pmc_split.sh <cfg> <outdir-under-gpurun_out>
set -u
CFG=${1:-0}
OUT=$GRAFT_REPO_ROOT/gpurun_out/${2:-pmc_split}
REPO=$GRAFT_REPO_ROOT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $grp -d $OUT/g$i -o p -- python $REPO/tools/gemm_split_pmc.py $CFG 6 > $OUT/g$i.log 2>&1
  echo "group $i ($grp) exit $?" >> $OUT/passes.log
done
python $REPO/tools/pmc_dump.py $(find $OUT -name 'p_results.db' | sort) > $OUT/summary.txt 2>&1
tail -5 $OUT/passes.log
