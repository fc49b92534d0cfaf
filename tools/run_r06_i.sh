#!/bin/bash
set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r06i
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in product lnrr; do
  for grp in FETCH_SIZE WRITE_SIZE; do
    if [ $v = product ]; then L=""; else L=tools/_tb/libt2h_$v.so; fi
    T2H_AB_LIB=$L timeout 200 rocprofv3 --pmc $grp -d $OUT/${v}_$grp -o p -- python $REPO/tools/ln_xcd_ab.py 8 > $OUT/${v}_$grp.log 2>&1
  done
  PMC_DUMP_FILTER=layernorm,gemm python $REPO/tools/pmc_dump.py $(find $OUT/${v}_FETCH_SIZE $OUT/${v}_WRITE_SIZE -name 'p_results.db' | sort) > $OUT/summary_$v.txt 2>&1
  echo "== $v"; cat $OUT/summary_$v.txt | cut -c1-160
done
rm -rf $OUT/*_FETCH_SIZE $OUT/*_WRITE_SIZE
