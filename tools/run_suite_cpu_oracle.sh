#!/bin/bash
# The whole GPU suite with every oracle comparison against the CPU execution of the oracle.  -> gpurun_out/round/
set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/round
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
S=$(date +%s)
T2H_TEST_ORACLE_DEVICE=cpu T2H_GPU_SUITE_BUDGET_S=0 timeout 1700 python -m pytest tests -q -m gpu --durations=15 > $OUT/gpu_suite_cpu_oracle.log 2>&1
echo "suite (oracle on the CPU) exit $? ($(( $(date +%s) - S )) s)" | tee -a $OUT/gpu_suite_cpu_oracle.log
tail -6 $OUT/gpu_suite_cpu_oracle.log
