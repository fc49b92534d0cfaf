#!/bin/bash
# Round-5 closing batch on the GPU box after the halo-staged convolution (a new kernel-source digest): the decode-side GPU
# tests, the driver's exact bench command, the kernel trace and the three --pmc passes of the parsing configuration (so
# that bench.py's roofline side data matches this tree), the per-shape convolution bench and the decode bench.
# Everything lands in gpurun_out/round/.  (The whole GPU suite takes 13 minutes: the driver runs it at round end.)
set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/round
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
S=$(date +%s)
timeout 420 python -m pytest -x -q \
    tests/test_gpu_conv_halo.py \
    "tests/test_gpu_conv_split.py::test_decode_with_split_convs_vs_reference_golden" \
    "tests/test_gpu_path.py::test_decode_vs_golden" "tests/test_gpu_path.py::test_end_to_end_vs_golden_with_reference_noise" \
    "tests/test_gpu_path.py::test_graft_entry_smoke_runs" "tests/test_gpu_path.py::test_api_surface_and_png_output" \
    "tests/test_gpu_edge_cases.py::test_batch_of_one_matches_oracle" "tests/test_gpu_edge_cases.py::test_decode_chunk_boundary_is_invisible" \
    "tests/test_gpu_edge_cases.py::test_decoder_is_resolution_agnostic" \
    "tests/test_gpu_edge_cases.py::test_stale_overflow_flag_is_not_this_runs_and_decode_checks_its_own" \
    "tests/test_gpu_edge_cases.py::test_decoder_attention_flash_fallback_matches_the_materialised_default" \
    "tests/test_gpu_configs.py::test_upscaled_hierarchy_1024x512" "tests/test_gpu_configs.py::test_upscaled_hierarchy_batch_of_8" \
    > $OUT/gpu_tests_decode_side.log 2>&1
echo "tests exit $? ($(( $(date +%s) - S )) s)" | tee -a $OUT/gpu_tests_decode_side.log
tail -6 $OUT/gpu_tests_decode_side.log
if [ "${ONLY_TESTS:-0}" = "1" ]; then exit 0; fi
S=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
echo "bench wall $(( $(date +%s) - S )) s" | tee $OUT/bench_driver_cmd.wall
cp gpurun_out/bench_detail.json $OUT/bench_driver_cmd_detail.json
tail -c 3000 $OUT/bench_driver_cmd.json
rm -rf $OUT/prof_parsing
rocprofv3 --kernel-trace --stats -d $OUT/prof_parsing -o p -- python bench.py --steps 1 --warmup 1 --config parsing \
    --no-cpu-baseline --no-exact-fp32 --no-other-configs --no-eager-leg --no-eager-gpu-baseline > $OUT/prof_parsing.log 2>&1
db=$(find $OUT/prof_parsing -name 'p_results.db' | head -1)
python tools/rocprof_summary.py $db $OUT/bench_parsing_kernel_stats.md > /dev/null
rm -rf $OUT/prof_parsing
bash tools/run_pmc_bench.sh parsing >> $OUT/pmc.log 2>&1
cp gpurun_out/pmc_summary_new*.md gpurun_out/pmc_summary_new*.json $OUT/ 2>/dev/null
python tools/conv_halo_bench.py 8 > $OUT/conv_halo_bench_b8.log 2>&1
python tools/decode_bench.py > $OUT/decode_bench.log 2>&1
tail -12 $OUT/conv_halo_bench_b8.log; cat $OUT/decode_bench.log | grep -v amdgpu
ls -la $OUT
