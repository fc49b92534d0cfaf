#!/bin/bash
# The whole GPU suite with per-test durations, as the driver runs it.  -> gpurun_out/round/gpu_suite.log
set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/round
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
S=$(date +%s)
timeout 1400 python -m pytest tests -q -m gpu --durations=40 > $OUT/gpu_suite.log 2>&1
echo "suite exit $? ($(( $(date +%s) - S )) s)" | tee -a $OUT/gpu_suite.log
tail -8 $OUT/gpu_suite.log
