#!/bin/bash
set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r06e
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
for b in 8 8 32 4; do timeout 200 python tools/mha_packed_ab.py $b >> $OUT/mha_ab.log 2>&1; done
grep -v amdgpu.ids $OUT/mha_ab.log
