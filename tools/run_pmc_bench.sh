#!/bin/bash
# The three rocprofv3 --pmc passes behind profiles/r04_pmc_summary.* (one counter group per pass, no
# tracing besides the kernel trace, as MI355X_MICROARCH.md prescribes) over a shortened bench step
# (--sample-steps 16: the per-launch counters of a kernel do not depend on how many rounds run).
set -u
REPO=$GRAFT_REPO_ROOT
ARGS="--steps 1 --warmup 1 --sample-steps 16 --no-cpu-baseline --no-exact-fp32 --no-other-configs --no-eager-leg"
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=pmc_$(echo $grp | cut -d' ' -f1)
  rm -rf $REPO/gpurun_out/$name
  timeout 300 rocprofv3 --pmc $grp -d $REPO/gpurun_out/$name -o p -- python $REPO/bench.py $ARGS > $REPO/gpurun_out/$name.log 2>&1
  echo "$name exit $?"
  find $REPO/gpurun_out/$name -name 'p_results.db' -exec mv {} $REPO/gpurun_out/$name/p_results.db \; 2>/dev/null
done
cd $REPO && python tools/pmc_summary.py gpurun_out gpurun_out/pmc_summary_new "bench.py $ARGS"
