#!/bin/bash
# After the absmax kernels' grid-stride rewrite (a new kernel-source digest): the measurement part of
# tools/run_final_r06.sh again, then the whole GPU suite on the final tree.
set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/round
mkdir -p $OUT
cd $REPO
NO_SUITE=1 bash tools/run_final_r06.sh > $OUT/measure.log 2>&1
tail -c 3300 $OUT/bench_driver_cmd.json
cd /tmp && export TMPDIR=/tmp
cd $REPO
S=$(date +%s)
timeout 1400 python -m pytest tests -q -m gpu --durations=40 > $OUT/gpu_suite.log 2>&1
echo "suite exit $? ($(( $(date +%s) - S )) s)" | tee -a $OUT/gpu_suite.log
tail -8 $OUT/gpu_suite.log
