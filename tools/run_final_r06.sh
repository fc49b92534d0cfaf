#!/bin/bash
# Round-6 closing batch on the GPU box (final kernel sources): the driver's exact bench command; rocprofv3 kernel traces
# of the same step for the three configurations; the three separate --pmc passes of the parsing and pose configurations
# (bench.py quotes them while the kernel-source digest matches); the per-kernel account of refine + decode; the whole
# GPU suite with per-test durations -- once as the driver runs it (oracle's convolutional stages as eager ROCm) and once
# with T2H_TEST_ORACLE_DEVICE=cpu (every comparison against the CPU execution of the oracle).  -> gpurun_out/round/
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
tail -c 3200 $OUT/bench_driver_cmd.json
trace() {  # name, bench arguments, extra env
  name=$1; shift
  rm -rf $OUT/prof_$name
  env "$@" rocprofv3 --kernel-trace --stats -d $OUT/prof_$name -o p -- python bench.py --steps 1 --warmup 1 \
      --no-cpu-baseline --no-exact-fp32 --no-other-configs --no-eager-leg --no-eager-gpu-baseline $TRACE_ARGS > $OUT/prof_$name.log 2>&1
  db=$(find $OUT/prof_$name -name 'p_results.db' | head -1)
  python tools/rocprof_summary.py $db $OUT/bench_${name}_kernel_stats.md > /dev/null
  rm -rf $OUT/prof_$name
}
for cfg in parsing pose hires; do TRACE_ARGS="--config $cfg" trace $cfg T2H_X8=1; done
for cfg in parsing pose; do
  bash tools/run_pmc_bench.sh $cfg >> $OUT/pmc.log 2>&1
done
cp gpurun_out/pmc_summary_new*.md gpurun_out/pmc_summary_new*.json $OUT/ 2>/dev/null
for up in 0 1; do
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace -d $OUT/dec$up -o p -- python $REPO/tools/decode_breakdown.py run 8 $up > $OUT/dec$up.log 2>&1
  cd $REPO
  python tools/decode_breakdown.py summarize $(find $OUT/dec$up -name 'p_results.db' | head -1) $OUT/decode_breakdown$([ $up = 1 ] && echo _hires).md 5 > /dev/null 2>&1
  rm -rf $OUT/dec$up
done
head -22 $OUT/decode_breakdown.md | cut -c1-150
if [ "${NO_SUITE:-0}" = "1" ]; then ls -la $OUT; exit 0; fi
S=$(date +%s)
timeout 1400 python -m pytest tests -q -m gpu --durations=40 > $OUT/gpu_suite.log 2>&1
echo "suite exit $? ($(( $(date +%s) - S )) s)" | tee -a $OUT/gpu_suite.log
tail -8 $OUT/gpu_suite.log
S=$(date +%s)
T2H_TEST_ORACLE_DEVICE=cpu T2H_GPU_SUITE_BUDGET_S=0 timeout 1700 python -m pytest tests -q -m gpu --durations=15 > $OUT/gpu_suite_cpu_oracle.log 2>&1
echo "suite (oracle on the CPU) exit $? ($(( $(date +%s) - S )) s)" | tee -a $OUT/gpu_suite_cpu_oracle.log
tail -6 $OUT/gpu_suite_cpu_oracle.log
ls -la $OUT
