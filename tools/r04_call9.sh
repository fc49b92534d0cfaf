#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c9; mkdir -p $O
python -m pytest tests/test_gpu_split.py -x -q -k "mha" > $O/tests.log 2>&1; tail -8 $O/tests.log
python tools/mha_bench.py 8 2>&1 | grep -v amdgpu; python tools/mha_bench.py 32 2>&1 | grep -v amdgpu
