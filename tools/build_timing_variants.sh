#!/bin/bash
# Debug builds of the split GEMM for tools/gemm_phase_timing.py (s_memrealtime / s_memtime stamps + ablation
# switches), made in the build container so that the GPU box only runs them: tools/_tb/<name>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_tb
build() {  # name, extra -D flags
  name=$1; shift
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-comment -Wno-pass-failed -Iinclude "$@" \
      text2human_amd/csrc/api.hip text2human_amd/csrc/gemm_split.hip -o tools/_tb/$name.so &
}
build none
build nogload -DT2H_SDBG_NOGLOAD
build nomma -DT2H_SDBG_NOMMA
build nofrag -DT2H_SDBG_NOFRAG
wait
ls -la tools/_tb
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-comment -Iinclude -DT2H_MHA_TIMING \
    text2human_amd/csrc/api.hip text2human_amd/csrc/attention.hip -o tools/_tb/mha_timing.so
ls -la tools/_tb
