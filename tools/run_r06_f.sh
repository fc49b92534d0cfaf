#!/bin/bash
set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r06f
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
for i in 1 2 3; do
  timeout 200 python tools/ln_xcd_ab.py 8 >> $OUT/ln_xcd_ab.log 2>&1
  T2H_AB_LIB=tools/_tb/libt2h_lnxcd.so timeout 200 python tools/ln_xcd_ab.py 8 >> $OUT/ln_xcd_ab.log 2>&1
done
timeout 200 python tools/ln_xcd_ab.py 32 >> $OUT/ln_xcd_ab.log 2>&1
T2H_AB_LIB=tools/_tb/libt2h_lnxcd.so timeout 200 python tools/ln_xcd_ab.py 32 >> $OUT/ln_xcd_ab.log 2>&1
grep -v amdgpu.ids $OUT/ln_xcd_ab.log
