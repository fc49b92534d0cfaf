#!/bin/bash
set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r06j
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
for i in 1 2 3 4 5; do
  timeout 300 python tools/sampler_lib_ab.py 8 >> $OUT/sampler_lib_ab.log 2>&1
  T2H_AB_LIB=tools/_tb/libt2h_lnrr.so timeout 300 python tools/sampler_lib_ab.py 8 >> $OUT/sampler_lib_ab.log 2>&1
done
grep "^product\|^tools" $OUT/sampler_lib_ab.log
