#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c4; mkdir -p $O
python -m pytest tests/test_gpu_split.py tests/test_gpu_edge_cases.py -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
python tools/sampler_gemm_bench.py 8 -1,8,10 7 > $O/gemm_b8.log 2>&1; grep cfg $O/gemm_b8.log
for ch in 1 2 4 1 2; do
  T2H_CHAINS=$ch python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-other-configs --no-eager-leg > $O/bench_ch$ch.json 2> $O/bench_ch$ch.err
  python - <<P
import json
d=json.load(open('$O/bench_ch$ch.json'))
print('chains $ch:', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms; sampler', round(d['stages']['sampler']['ms_per_step'],1), 'ms_per_round', round(d['stages']['sampler']['ms_per_round'],3), d['launch_mode'][:12])
P
done
