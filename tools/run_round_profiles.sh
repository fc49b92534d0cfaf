#!/bin/bash
# Round-end measurement batch on the GPU box: bench lines of the three configs, rocprofv3 kernel
# traces of the same commands, and the three separate --pmc passes (tools/run_pmc_bench.sh).
set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/round
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
python bench.py --steps 5 --warmup 2 > $OUT/bench_parsing.json 2> $OUT/bench_parsing.err
python bench.py --config pose --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 > $OUT/bench_pose.json 2> $OUT/bench_pose.err
python bench.py --config hires --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 > $OUT/bench_hires.json 2> $OUT/bench_hires.err
for cfg in parsing pose hires; do
  rm -rf $OUT/prof_$cfg
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$cfg -o p -- python bench.py --config $cfg --steps 1 --warmup 1 \
      --no-cpu-baseline --no-exact-fp32 > $OUT/prof_$cfg.log 2>&1
  db=$(find $OUT/prof_$cfg -name 'p_results.db' | head -1)
  python tools/rocprof_summary.py $db $OUT/${cfg}_kernel_stats.md > /dev/null
  rm -rf $OUT/prof_$cfg
done
bash tools/run_pmc_bench.sh > $OUT/pmc.log 2>&1
cp gpurun_out/pmc_summary_new.md gpurun_out/pmc_summary_new.json $OUT/ 2>/dev/null
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_SQ_VALU_MFMA_BUSY_CYCLES
ls -la $OUT
