#!/bin/bash
# usage: tools/build_variant.sh <name> "<extra hipcc flags>" [source.hip ...]  ->  build_ab/<name>.so
# Rebuilds the named sources (default gemm_nt_v3.hip) with the extra flags and links them with the shipped objects.
set -e
name=$1; flags=$2; shift 2 || true
srcs=${@:-gemm_nt_v3.hip}
cd "$(dirname "$0")/../dino_amd/csrc"
make -s >/dev/null 2>&1 || make
mkdir -p ../../build_ab/obj_$name
objs=""
for o in build/*.o; do
  b=$(basename $o .o); skip=0
  for s in $srcs; do [ "$b.hip" = "$s" ] && skip=1; done
  [ $skip = 0 ] && objs="$objs $o"
done
for s in $srcs; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function $flags -c $s -o ../../build_ab/obj_$name/${s%.hip}.o 2>&1 | grep -E "error|spill" || true
  objs="$objs ../../build_ab/obj_$name/${s%.hip}.o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_ab/$name.so $objs
echo "built build_ab/$name.so ($flags)"
