#!/bin/bash
set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r06c
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
timeout 900 python tools/oracle_device_check.py 8 > $OUT/oracle_device_check.log 2>&1; echo "oracle check exit $?"
grep -v amdgpu.ids $OUT/oracle_device_check.log | tail -8
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/dec -o p -- python $REPO/tools/decode_breakdown.py run 8 0 > $OUT/dec.log 2>&1
cd $REPO
python tools/decode_breakdown.py summarize $(find $OUT/dec -name 'p_results.db' | head -1) $OUT/decode_breakdown.md 5 > $OUT/decode_breakdown.log 2>&1
head -60 $OUT/decode_breakdown.md | cut -c1-160
rm -rf $OUT/dec
