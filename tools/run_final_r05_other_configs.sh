#!/bin/bash
# Kernel traces of the pose and 1024x512 configurations and the pose --pmc passes on the final round-5 sources (the
# parsing configuration's are made by tools/run_final_r05_measure.sh).  Output: gpurun_out/round/.
set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/round
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
for cfg in pose hires; do
  rm -rf $OUT/prof_$cfg
  timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_$cfg -o p -- python bench.py --steps 1 --warmup 1 --config $cfg \
      --no-cpu-baseline --no-exact-fp32 --no-other-configs --no-eager-leg --no-eager-gpu-baseline > $OUT/prof_$cfg.log 2>&1
  echo "$cfg trace exit $?"
  db=$(find $OUT/prof_$cfg -name 'p_results.db' | head -1)
  python tools/rocprof_summary.py $db $OUT/bench_${cfg}_kernel_stats.md > /dev/null
  rm -rf $OUT/prof_$cfg
done
timeout 280 bash tools/run_pmc_bench.sh pose >> $OUT/pmc_pose.log 2>&1
cp gpurun_out/pmc_summary_new_pose.md gpurun_out/pmc_summary_new_pose.json $OUT/ 2>/dev/null
ls -la $OUT | grep -i "pose\|hires"
