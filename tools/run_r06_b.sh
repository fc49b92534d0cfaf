#!/bin/bash
# (ran on the tree of commit 7353688 where the A/B switch T2H_CONV_SPLITK still existed; it was removed with the round's
# closing commit -- ops.conv3x3(ksplit=1) is the single-pass form -- so the second pass below now repeats the first)
# Round-6 batch B: where the oracle's time goes (CPU vs eager ROCm), the decode account with / without the exact-fp32
# convolutions' split over K, and the parity tests the split touches.  -> gpurun_out/r06b/
set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r06b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
timeout 600 python tools/oracle_device_check.py 8 > $OUT/oracle_device_check.log 2>&1; echo "oracle check exit $?"
grep -v amdgpu.ids $OUT/oracle_device_check.log | tail -8
for sk in 1 0; do
  cd /tmp
  T2H_CONV_SPLITK=$sk timeout 300 rocprofv3 --kernel-trace -d $OUT/dec_sk$sk -o p -- python $REPO/tools/decode_breakdown.py run 8 0 > $OUT/dec_sk$sk.log 2>&1
  cd $REPO
  python tools/decode_breakdown.py summarize $(find $OUT/dec_sk$sk -name 'p_results.db' | head -1) $OUT/decode_breakdown_splitk$sk.md 5 > $OUT/decode_breakdown_sk$sk.log 2>&1
  head -24 $OUT/decode_breakdown_splitk$sk.md | cut -c1-160
  rm -rf $OUT/dec_sk$sk
done
S=$(date +%s)
T2H_GPU_SUITE_BUDGET_S=0 timeout 1200 python -m pytest -x -q --durations=25 tests/test_gpu_kernels.py tests/test_gpu_path.py tests/test_gpu_vs_reference.py \
   tests/test_gpu_edge_cases.py tests/test_gpu_configs.py "tests/test_gpu_bench_parity.py::test_bench_config_parity" tests/test_gpu_ui_hooks.py tests/test_gpu_encode.py \
   > $OUT/tests.log 2>&1
echo "tests exit $? ($(( $(date +%s) - S )) s)"
tail -40 $OUT/tests.log
