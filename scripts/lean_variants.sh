#!/bin/bash
# Builds variant libraries of the lean kernel (admm_lean.hip.h tuning macros) next to the product library:
#   lib/variants/libtinympc_hip_<tag>.so = the product's objects with linst_* recompiled under -D<macros>
# usage: scripts/lean_variants.sh tag1:"-DTMPC_LEAN_WAVES=1" tag2:"-DTMPC_LEAN_D64=1 -DTMPC_LEAN_WAVES=1" ...
set -e
cd "$(dirname "$0")/../tinympc-julia_amd/csrc"
mkdir -p ../lib/variants build/variants
for spec in "$@"; do
  tag="${spec%%:*}"; flags="${spec#*:}"
  objs=""
  for f in linst_*.hip; do
    o="build/variants/${f%.hip}_$tag.o"
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-honor-nans -fno-slp-vectorize $flags -c "$f" -o "$o"
    objs="$objs $o"
  done
  base=$(ls build/*.o | grep -v "/linst_")
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "../lib/variants/libtinympc_hip_$tag.so" $base $objs
  echo "built variant $tag ($flags)"
done
