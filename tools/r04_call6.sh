#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c6; mkdir -p $O
T2H_TIMING_SO=tools/_tb/mha_timing.so python tools/mha_phase_timing.py 8 2>&1 | grep -v amdgpu.ids > $O/mha_phase.log
T2H_TIMING_SO=tools/_tb/mha_timing.so python tools/mha_phase_timing.py 32 2>&1 | grep -v amdgpu.ids >> $O/mha_phase.log
cat $O/mha_phase.log
which rocm-smi amd-smi; rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -30 > $O/smi_idle.log; cat $O/smi_idle.log
( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr '\n' ' '; echo; sleep 0.25; done ) > $O/smi_during.log 2>&1 &
SMI=$!
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-other-configs --no-eager-leg > $O/bench.json 2> $O/bench.err
kill $SMI
python - <<P
import json
d=json.load(open('$O/bench.json')); print('bench', d['value'], d['ms_per_step'])
P
tail -25 $O/smi_during.log
