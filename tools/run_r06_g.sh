#!/bin/bash
set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r06g
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
for i in 1 2; do
  for v in "" lnrr lnplain mhaplain gemmplain; do
    if [ -z "$v" ]; then timeout 300 python tools/sampler_lib_ab.py 8 >> $OUT/sampler_lib_ab.log 2>&1
    else T2H_AB_LIB=tools/_tb/libt2h_$v.so timeout 300 python tools/sampler_lib_ab.py 8 >> $OUT/sampler_lib_ab.log 2>&1; fi
  done
done
grep -v "amdgpu.ids\|Warning\|warn" $OUT/sampler_lib_ab.log
