#!/bin/bash
# The three rocprofv3 --pmc passes behind profiles/r04_pmc_summary*.{md,json} (one counter group per pass, no
# tracing besides the kernel trace, as MI355X_MICROARCH.md prescribes) over a shortened bench step of ONE
# configuration (--sample-steps 16: the per-launch counters of a kernel do not depend on how many rounds run;
# T2H_GRAPH=0: individual launches, so that every dispatch is attributed as in the kernel trace).
#   bash tools/run_pmc_bench.sh [parsing|pose|hires]
set -u
CFG=${1:-parsing}
REPO=$GRAFT_REPO_ROOT
ARGS="--config $CFG --steps 1 --warmup 1 --sample-steps 16 --no-cpu-baseline --no-exact-fp32 --no-other-configs --no-eager-leg --no-eager-gpu-baseline"
cd /tmp && export TMPDIR=/tmp
export T2H_GRAPH=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=pmc_$(echo $grp | cut -d' ' -f1)
  rm -rf $REPO/gpurun_out/$name
  timeout 300 rocprofv3 --pmc $grp -d $REPO/gpurun_out/$name -o p -- python $REPO/bench.py $ARGS > $REPO/gpurun_out/$name.log 2>&1
  echo "$CFG $name exit $?"
  find $REPO/gpurun_out/$name -name 'p_results.db' -exec mv {} $REPO/gpurun_out/$name/p_results.db \; 2>/dev/null
done
TAG=""; [ "$CFG" != "parsing" ] && TAG="_$CFG"
cd $REPO && python tools/pmc_summary.py gpurun_out gpurun_out/pmc_summary_new$TAG "T2H_GRAPH=0 bench.py $ARGS"
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_SQ_VALU_MFMA_BUSY_CYCLES
