#!/bin/bash
# cache policy of the LDS-DMA tile loads of config 8 (timing build): default, nt, sc1, sc0 sc1
export T2H_TIMING_SHAPES=${T2H_TIMING_SHAPES:-fc1,qkv_nov}
for pol in "-" "nt" "sc1" "sc0 sc1" "sc0" "-"; do  # "-" = default policy
  echo "policy: '$pol'"
  T2H_DMA_POLICY="$pol" timeout 120 python tools/gemm_phase_timing.py 8 2>&1 | grep -v "amdgpu.ids\|^defs" | cut -c1-170
done
