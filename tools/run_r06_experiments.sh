#!/bin/bash
# Round-6 measurement batch: the split-K experiment on proj / fc2, the packed-fp32 A/B
# and the counter passes of the attention kernel, the per-kernel account of refine + decode.  -> gpurun_out/r06x/
set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r06x
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
# --- item 4: proj / fc2 on the ping-pong loop with a split over K across workgroups (partial fp32 tiles)
T2H_TIMING_X8=1 T2H_TIMING_SHAPES=proj,fc2,fc1 timeout 300 python tools/gemm_phase_timing.py 6,0,11,11/2,11/4,8,8/2,8/4 8 \
    > $OUT/fc2_128x128_splitk_b8.log 2>&1
echo "splitk b8 exit $?"
T2H_TIMING_X8=1 T2H_TIMING_SHAPES=proj,fc2 timeout 300 python tools/gemm_phase_timing.py 6,11,11/2,8,8/2 32 \
    > $OUT/fc2_128x128_splitk_b32.log 2>&1
echo "splitk b32 exit $?"
cat $OUT/fc2_128x128_splitk_b8.log | cut -c1-400
# --- item 5: attention, packed-fp32 softmax A/B + counters
for b in 8 32; do timeout 200 python tools/mha_packed_ab.py $b >> $OUT/mha_packed_ab.log 2>&1; done
cat $OUT/mha_packed_ab.log
bash tools/run_pmc_mha.sh r06x/pmc_mha > $OUT/pmc_mha_passes.log 2>&1
cat $OUT/pmc_mha_passes.log | tail -12
head -60 $OUT/pmc_mha/summary.txt
# --- item 6c: per-kernel account of refine + decode
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/dec_b8 -o p -- python $REPO/tools/decode_breakdown.py run 8 0 > $OUT/dec_b8.log 2>&1
echo "decode trace exit $?"
cd $REPO
python tools/decode_breakdown.py summarize $(find $OUT/dec_b8 -name 'p_results.db' | head -1) $OUT/decode_breakdown.md 5 > $OUT/decode_breakdown.log 2>&1
cat $OUT/decode_breakdown.md
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/dec_hires -o p -- python $REPO/tools/decode_breakdown.py run 8 1 > $OUT/dec_hires.log 2>&1
cd $REPO
python tools/decode_breakdown.py summarize $(find $OUT/dec_hires -name 'p_results.db' | head -1) $OUT/decode_breakdown_hires.md 5 > $OUT/decode_breakdown_hires.log 2>&1
head -40 $OUT/decode_breakdown_hires.md
rm -rf $OUT/dec_b8 $OUT/dec_hires
