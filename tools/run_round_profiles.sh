#!/bin/bash
# Round-end measurement batch on the GPU box (run through gpurun from the repo root):
#   the driver's exact bench command, rocprofv3 kernel traces of the same step for the three configurations (full 256
#   sampling steps, the default path: every round one hipGraph replay) and of the parsing configuration with T2H_X8=0
#   (the A/B of DESIGN.md 4.7), the three separate --pmc passes of the parsing and pose configurations, the
#   micro-benchmarks behind DESIGN.md's tables.  Everything lands in gpurun_out/round/; copy what is to be judged to
#   profiles/r05_* (tools/rocprof_summary.py / tools/pmc_summary.py stamp the kernel-source digest).
set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/round
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
S=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
echo "bench wall $(( $(date +%s) - S )) s" > $OUT/bench_driver_cmd.wall
cp gpurun_out/bench_detail.json $OUT/bench_driver_cmd_detail.json
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
TRACE_ARGS="--config parsing" trace parsing_fp16_planes T2H_X8=0
for cfg in parsing pose; do
  bash tools/run_pmc_bench.sh $cfg >> $OUT/pmc.log 2>&1
done
cp gpurun_out/pmc_summary_new*.md gpurun_out/pmc_summary_new*.json $OUT/ 2>/dev/null
python tools/x8_gemm_bench.py 8 > $OUT/x8_gemm_bench_b8.log 2>&1
python tools/x8_gemm_bench.py 32 > $OUT/x8_gemm_bench_b32.log 2>&1
T2H_TIMING_X8=1 T2H_TIMING_X8_OUT=0 python tools/gemm_phase_timing.py 6,8,10 8 > $OUT/gemm_phase_timing_x8_b8.log 2>&1
python tools/gemm_phase_timing.py 6,8,10 8 > $OUT/gemm_phase_timing_b8.log 2>&1
python tools/mha_forms_bench.py > $OUT/mha_all_keys.log 2>&1
python tools/decode_bench.py > $OUT/decode_bench.log 2>&1
ls -la $OUT
