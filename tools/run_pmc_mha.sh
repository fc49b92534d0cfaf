#!/bin/bash
# rocprofv3 --pmc passes (one small counter group per pass, kernel trace only) over tools/mha_pmc.py: where the 63 %
# idle matrix pipe of mha_split_pipe_kernel<2> goes.  A group with a counter this box does not
# offer fails alone.  -> gpurun_out/<out>/summary.txt (tools/pmc_dump.py)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_mha}
REPO=$GRAFT_REPO_ROOT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_TRANS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INST_LEVEL_LDS" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_INSTS_VALU_TRANS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp -d $OUT/g$i -o p -- python $REPO/tools/mha_pmc.py 8 > $OUT/g$i.log 2>&1
  echo "group $i ($grp) exit $?" >> $OUT/passes.log
done
PMC_DUMP_FILTER=mha python $REPO/tools/pmc_dump.py $(find $OUT -name 'p_results.db' | sort) > $OUT/summary.txt 2>&1
cat $OUT/passes.log
rm -rf $OUT/g*/
