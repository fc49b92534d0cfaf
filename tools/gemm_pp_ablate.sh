#!/bin/bash
# ablation of the ping-pong LDS-DMA main loop (config 8) next to the register-staged one (config 4):
# the timing build with pieces switched off (T2H_SDBG_NOGLOAD / NOPUT / NOFRAG / NOMMA / NOBAR / HOTLOAD /
# HALFDMA / DMAFIRST, comma-separated sets in the list below)
export T2H_TIMING_SHAPES=${T2H_TIMING_SHAPES:-fc1}
for d in "" T2H_SDBG_NOGLOAD T2H_SDBG_NOFRAG T2H_SDBG_NOMMA T2H_SDBG_HOTLOAD T2H_SDBG_HALFDMA; do
  T2H_TIMING_DEFS=$d timeout 120 python tools/gemm_phase_timing.py ${CFGS:-4,8} 2>&1 | grep -v amdgpu.ids | cut -c1-170
done
