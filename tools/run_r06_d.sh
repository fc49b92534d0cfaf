#!/bin/bash
set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r06d
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
timeout 600 python tools/halo_stagger_ab.py 8 0 > $OUT/halo_stagger_ab.log 2>&1; echo "stagger exit $?"
timeout 600 python tools/halo_stagger_ab.py 8 1 >> $OUT/halo_stagger_ab.log 2>&1; echo "stagger hires exit $?"
grep -v amdgpu.ids $OUT/halo_stagger_ab.log | tail -20
S=$(date +%s)
T2H_GPU_SUITE_BUDGET_S=0 timeout 1500 python -m pytest tests -q -m gpu --durations=60 > $OUT/gpu_suite.log 2>&1
echo "suite exit $? ($(( $(date +%s) - S )) s)" | tee -a $OUT/gpu_suite.log
tail -80 $OUT/gpu_suite.log
