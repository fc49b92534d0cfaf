#!/bin/bash
# round-4 GPU batch 2: W4 split GEMM correctness + A/B, pose B=32 parity rerun, edge cases, bench with the graph default
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c2; mkdir -p $O
python -m pytest tests/test_gpu_split.py -x -q > $O/split_tests.log 2>&1; tail -3 $O/split_tests.log
python tools/sampler_gemm_bench.py 8 -1,8,13,10,14 7 > $O/gemm_b8.log 2>&1; cat $O/gemm_b8.log
python tools/sampler_gemm_bench.py 32 -1,8,13 5 > $O/gemm_b32.log 2>&1; cat $O/gemm_b32.log
T2H_TIMING_SHAPES=fc1 python tools/gemm_phase_timing.py 8,13 8 > $O/phase_fc1.log 2>&1; cat $O/phase_fc1.log
T2H_TIMING_SHAPES=qkv_nov python tools/gemm_phase_timing.py 10,14 8 > $O/phase_qkv.log 2>&1; cat $O/phase_qkv.log
python -m pytest tests/test_gpu_configs.py::test_sample_from_pose_batch_32_teacher_forced tests/test_gpu_edge_cases.py -x -q -s > $O/tests2.log 2>&1; tail -5 $O/tests2.log; grep "pose B=32" $O/tests2.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; tail -3 $O/bench.err
