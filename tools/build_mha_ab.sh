#!/bin/bash
# The two builds of csrc/attention.hip for tools/mha_packed_ab.py (made in the build container; the GPU box only runs them)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_tb
for v in 0 1; do
  name=$([ $v = 0 ] && echo base || echo packed)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-comment -Iinclude -DT2H_MHA_PACKED=$v \
      text2human_amd/csrc/api.hip text2human_amd/csrc/attention.hip -o tools/_tb/mha_$name.so &
done
wait
ls -la tools/_tb/mha_*.so
