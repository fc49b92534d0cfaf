#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c10; mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=8 > $O/gputests.log 2>&1; tail -14 $O/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python tools/gemm_phase_timing.py 6,8,10 8 2>&1 | grep -v "amdgpu\|warning\|^ \|^$" > $O/gemm_phase_timing_b8.log; cat $O/gemm_phase_timing_b8.log
python tools/decode_bench.py > $O/decode_bench.log 2>&1; tail -12 $O/decode_bench.log
