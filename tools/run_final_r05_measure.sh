#!/bin/bash
# The measurement part of tools/run_final_r05.sh alone (after a comment-only change of a kernel source: a new digest):
# the driver's exact bench command, the kernel trace and the three --pmc passes of the parsing configuration.
set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/round
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
S=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
echo "bench wall $(( $(date +%s) - S )) s" | tee $OUT/bench_driver_cmd.wall
cp gpurun_out/bench_detail.json $OUT/bench_driver_cmd_detail.json
tail -c 2900 $OUT/bench_driver_cmd.json
rm -rf $OUT/prof_parsing
rocprofv3 --kernel-trace --stats -d $OUT/prof_parsing -o p -- python bench.py --steps 1 --warmup 1 --config parsing \
    --no-cpu-baseline --no-exact-fp32 --no-other-configs --no-eager-leg --no-eager-gpu-baseline > $OUT/prof_parsing.log 2>&1
db=$(find $OUT/prof_parsing -name 'p_results.db' | head -1)
python tools/rocprof_summary.py $db $OUT/bench_parsing_kernel_stats.md > /dev/null
rm -rf $OUT/prof_parsing
bash tools/run_pmc_bench.sh parsing >> $OUT/pmc.log 2>&1
cp gpurun_out/pmc_summary_new*.md gpurun_out/pmc_summary_new*.json $OUT/ 2>/dev/null
ls -la $OUT
