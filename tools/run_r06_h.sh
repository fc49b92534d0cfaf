#!/bin/bash
set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r06h
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
for i in 1 2 3; do
  for v in "" lnrr lnrpb4 lnrpb16; do
    if [ -z "$v" ]; then timeout 200 python tools/ln_xcd_ab.py 8 >> $OUT/ln_ab.log 2>&1
    else T2H_AB_LIB=tools/_tb/libt2h_$v.so timeout 200 python tools/ln_xcd_ab.py 8 >> $OUT/ln_ab.log 2>&1; fi
  done
done
grep -v amdgpu.ids $OUT/ln_ab.log
