#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c3; mkdir -p $O
for v in none nogload nomma nofrag; do
  T2H_TIMING_SO=tools/_tb/$v.so T2H_TIMING_SHAPES=fc1 python tools/gemm_phase_timing.py 8,13 8 2>/dev/null | grep -v amdgpu.ids | sed "s/^/$v: /"
done > $O/phase_ablate.log 2>&1
T2H_TIMING_ZERO=1 T2H_TIMING_SO=tools/_tb/none.so T2H_TIMING_SHAPES=fc1 python tools/gemm_phase_timing.py 8,13 8 2>/dev/null | grep -v amdgpu.ids | sed "s/^/zero-data: /" >> $O/phase_ablate.log
T2H_TIMING_SO=tools/_tb/none.so T2H_TIMING_SHAPES=fc1,qkv_nov,proj,fc2 python tools/gemm_phase_timing.py 6,8,10 8 2>/dev/null | grep -v amdgpu.ids | sed "s/^/all: /" >> $O/phase_ablate.log
T2H_TIMING_SO=tools/_tb/none.so T2H_TIMING_SHAPES=fc1,fc2 python tools/gemm_phase_timing.py 8 32 2>/dev/null | grep -v amdgpu.ids | sed "s/^/B32: /" >> $O/phase_ablate.log
cat $O/phase_ablate.log
python tools/mha_bench.py > $O/mha_bench.log 2>&1; tail -12 $O/mha_bench.log
python -m pytest tests/test_gpu_configs.py::test_sample_from_pose_batch_32_teacher_forced tests/test_gpu_edge_cases.py -x -q -s > $O/tests.log 2>&1; tail -4 $O/tests.log; grep "pose B=32" $O/tests.log
