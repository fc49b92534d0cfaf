#!/bin/bash
# Variant libraries for tools/ln_xcd_ab.py / tools/sampler_lib_ab.py: the product objects with ONE kernel file rebuilt
# under an A/B macro.  tools/_tb/libt2h_<name>.so
#   lnrr      : csrc/norm.hip with -DT2H_LN_XCD=0 (LayerNorm rows handed out round-robin: the mapping until round 6)
set -e
cd "$(dirname "$0")/.."
python -m text2human_amd.build
mkdir -p tools/_tb
variant() {  # name, source file (without .hip), flags
  name=$1; src=$2; shift 2
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment "$@" -c text2human_amd/csrc/$src.hip -o tools/_tb/${src}_$name.o
  objs=$(ls text2human_amd/csrc/build/*.o | grep -v "/$src.o")
  hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/_tb/${src}_$name.o -o tools/_tb/libt2h_$name.so
}
variant lnrr norm -DT2H_LN_XCD=0 &
wait
ls -la tools/_tb/libt2h_*.so
