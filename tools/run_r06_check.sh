#!/bin/bash
# Round-6 GPU batch: the x8 purity / outlier tests first, then the whole GPU suite with per-test durations, then the
# driver's bench command.  Everything lands in gpurun_out/r06/.
set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
S=$(date +%s)
timeout 900 python -m pytest -x -q tests/test_gpu_x8.py "tests/test_gpu_bench_parity.py::test_outlier_channels" \
    --durations=15 > $OUT/x8_tests.log 2>&1
echo "x8 tests exit $? ($(( $(date +%s) - S )) s)" | tee -a $OUT/x8_tests.log
tail -30 $OUT/x8_tests.log
cp gpurun_out/parity_outliers.json $OUT/ 2>/dev/null
if [ "${ONLY_X8:-0}" = "1" ]; then exit 0; fi
S=$(date +%s)
T2H_GPU_SUITE_BUDGET_S=0 timeout 1500 python -m pytest tests -q -m gpu --durations=70 > $OUT/gpu_suite.log 2>&1
echo "suite exit $? ($(( $(date +%s) - S )) s)" | tee -a $OUT/gpu_suite.log
tail -90 $OUT/gpu_suite.log
S=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
echo "bench exit $? wall $(( $(date +%s) - S )) s" | tee $OUT/bench_driver_cmd.wall
cp gpurun_out/bench_detail.json $OUT/bench_driver_cmd_detail.json
tail -c 3200 $OUT/bench_driver_cmd.json
