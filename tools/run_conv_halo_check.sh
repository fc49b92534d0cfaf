#!/bin/bash
# GPU batch for the halo-staged convolution: its parity tests, the per-shape bench against the two-kernel path, the
# decode stage with the kernel off / on / everywhere, and three separate --pmc passes (FETCH_SIZE, WRITE_SIZE, matrix
# pipe busy) over the two largest shapes.  Output: gpurun_out/halo/.
set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/halo
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
timeout 420 python -m pytest tests/test_gpu_conv_halo.py -x -q > $OUT/tests.log 2>&1
echo "tests exit $?" | tee -a $OUT/tests.log
tail -15 $OUT/tests.log
timeout 300 python tools/conv_halo_bench.py 8 > $OUT/conv_halo_bench_b8.log 2>&1
echo "bench exit $?"
cat $OUT/conv_halo_bench_b8.log
if [ "${HALO_PMC:-1}" = "1" ]; then
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    name=pmc_$(echo $grp | cut -d' ' -f1)
    rm -rf $OUT/$name
    CONV_HALO_BENCH_SHAPES=2 timeout 200 rocprofv3 --pmc $grp -d $OUT/$name -o p -- python tools/conv_halo_bench.py 8 > $OUT/$name.log 2>&1
    echo "$name exit $?"
  done
  PMC_DUMP_FILTER=conv,gn_apply python tools/pmc_dump.py $(find $OUT -name 'p_results.db') > $OUT/pmc_dump.txt 2>&1
  cat $OUT/pmc_dump.txt
  rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES
fi
