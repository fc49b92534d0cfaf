#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c7; mkdir -p $O
python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_kernels.py tests/test_gpu_path.py -x -q > $O/tests.log 2>&1; tail -6 $O/tests.log
for sh in 1 0 1 0; do
  T2H_SHRINK_BATCH=$sh python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-eager-leg --other-steps 2 > $O/bench_sh$sh.json 2> $O/bench_sh$sh.err
  python - <<P
import json
d=json.load(open('$O/bench_sh$sh.json'))
s=d['stages']['sampler']
print('shrink $sh:', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms; sampler', round(s['ms_per_step'],1), 'evaluated', s['sample_steps_evaluated'], 'needed', s['sample_steps_needed'], '| b32', round(d['other_configs']['parsing_b32']['value'],2), 'pose', round(d['other_configs']['pose']['value'],2), 'hires', round(d['other_configs']['hires']['value'],2))
P
done
