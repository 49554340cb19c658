#!/bin/bash
# attention kernels at the headline shapes: build_ab/prev.so against build_ab/new.so (kernel-level)
set -e
L=dino_amd/lib/libdino_hip.so
cp $L /tmp/keep.so
for v in ${VARIANTS:-prev new}; do
  cp build_ab/$v.so $L
  TAG=$v timeout -k 10 100 python tools/bench_attn.py
done
cp /tmp/keep.so $L
