#!/bin/bash
# builds libptx_hip.so variants with extra -D flags into tools/dbg/variants/ (parallel); usage: build_variants.sh name1 "flags1" name2 "flags2" ...
cd "$(dirname "$0")/../../path-tracing_amd/csrc"
OUT=../../tools/dbg/variants
while [ $# -gt 1 ]; do
  n=$1; f=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function $f -shared -o $OUT/libptx_$n.so ptx_hip.hip &
done
wait
ls -la $OUT
