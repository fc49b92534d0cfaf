#!/bin/bash
# The driver's bench command three times in a row on ONE box: run-to-run spread (next to the box-to-box spread of the
# round's closing runs).  -> gpurun_out/round/bench_repeat.log
set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/round
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
: > $OUT/bench_repeat.log
for i in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2> /dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({k:d[k] for k in ('value','ms_per_step','ms_per_step_median','stages_ms')} | {'roofline_frac': d['roofline']['frac'], 'traffic': d['roofline'].get('traffic'), 'mfma_util_pmc': d['roofline'].get('mfma_util_pmc'), 'decode_ms_per_image': d['decode']['ms_per_image'], 'fp16_planes': d['fp16_planes_path']['value'], 'other': {k:v['value'] for k,v in d['other_configs'].items()}}))
" >> $OUT/bench_repeat.log
done
cat $OUT/bench_repeat.log
