#!/bin/bash
# ablation of the ping-pong main loop (cfg 7) next to the plain one (cfg 4), fc1 + qkv shapes
export T2H_TIMING_SHAPES=${T2H_TIMING_SHAPES:-fc1}
for d in "" T2H_SDBG_DMAFIRST; do
  T2H_TIMING_DEFS=$d timeout 120 python tools/gemm_phase_timing.py ${CFGS:-4,7} 2>&1 | grep -v amdgpu.ids | cut -c1-170
done
