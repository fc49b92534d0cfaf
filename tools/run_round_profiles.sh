#!/bin/bash
# Round-end measurement batch on the GPU box (run through gpurun from the repo root):
#   the bench line, rocprofv3 kernel traces of the same command for the three configurations (full 256 sampling
#   steps, the default path: every round one hipGraph replay), the three separate --pmc passes of EACH configuration,
#   the micro-benchmarks behind DESIGN.md's tables.  Everything lands in gpurun_out/round/; copy what is to be judged
#   to profiles/r04_* (tools/rocprof_summary.py / tools/pmc_summary.py stamp the kernel-source digest).
set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/round
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
python bench.py --steps 5 --warmup 2 > $OUT/bench_parsing.json 2> $OUT/bench_parsing.err
for cfg in parsing pose hires; do
  rm -rf $OUT/prof_$cfg
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$cfg -o p -- python bench.py --config $cfg --steps 1 --warmup 1 \
      --no-cpu-baseline --no-exact-fp32 --no-other-configs --no-eager-leg > $OUT/prof_$cfg.log 2>&1
  db=$(find $OUT/prof_$cfg -name 'p_results.db' | head -1)
  python tools/rocprof_summary.py $db $OUT/bench_${cfg}_kernel_stats.md > /dev/null
  rm -rf $OUT/prof_$cfg
done
for cfg in parsing pose hires; do
  bash tools/run_pmc_bench.sh $cfg >> $OUT/pmc.log 2>&1
done
cp gpurun_out/pmc_summary_new*.md gpurun_out/pmc_summary_new*.json $OUT/ 2>/dev/null
python tools/sampler_gemm_bench.py 8 -1,8,10 7 > $OUT/sampler_gemm_bench_b8.log 2>&1
python tools/mha_bench.py > $OUT/mha_bench.log 2>&1
python tools/gemm_phase_timing.py 6,8,10 8 > $OUT/gemm_phase_timing_b8.log 2>&1
T2H_TIMING_SO=tools/_tb/mha_timing.so python tools/mha_phase_timing.py 8 > $OUT/mha_phase_timing_b8.log 2>&1
ls -la $OUT
