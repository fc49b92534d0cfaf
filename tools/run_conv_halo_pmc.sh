#!/bin/bash
# Stall accounting of the halo-staged convolution against the two-kernel path on the decoders' largest shapes:
# per-shape bench (first $SHAPES shapes, both main-loop variants), then separate --pmc passes of SQ wait / LDS counters
# (the bench's last variant -- the default -- and variant 0 both appear, by template arguments).  Output: gpurun_out/halo/.
set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/halo
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
if [ "${HALO_TESTS:-0}" = "1" ]; then
  timeout 420 python -m pytest tests/test_gpu_conv_halo.py -x -q > $OUT/tests.log 2>&1
  echo "tests exit $?"; tail -4 $OUT/tests.log
fi
export CONV_HALO_BENCH_SHAPES=${SHAPES:-4}
timeout 200 python tools/conv_halo_bench.py 8 > $OUT/conv_halo_bench_top.log 2>&1
echo "bench exit $?"; cat $OUT/conv_halo_bench_top.log
export CONV_HALO_BENCH_SHAPES=1
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rm -rf $OUT/sq$i
  timeout 150 rocprofv3 --pmc $grp -d $OUT/sq$i -o p -- python tools/conv_halo_bench.py 8 > $OUT/sq$i.log 2>&1
  echo "sq$i exit $?"
done
PMC_DUMP_FILTER=conv_halo python tools/pmc_dump.py $(find $OUT -name 'p_results.db') > $OUT/sq_dump.txt 2>&1
cat $OUT/sq_dump.txt
rm -rf $OUT/sq1 $OUT/sq2 $OUT/sq3
