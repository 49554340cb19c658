#!/bin/bash
# GPU box: ViT-B/16 and ViT-L/14 steps with two builds of the library (build_ab/prev.so, build_ab/new.so), alternating
set -e
L=dino_amd/lib/libdino_hip.so
cp $L /tmp/keep.so
trap "cp /tmp/keep.so $L" EXIT
for cfg in ${CONFIGS:-vitb16 vitl14}; do
  for v in prev new prev new; do
    cp build_ab/$v.so $L
    echo -n "$v "; timeout -k 10 200 python tools/bench_other_configs.py $cfg 2>/dev/null | tail -1 | cut -c1-200
  done
done
