#!/bin/bash
# same-box timing of NT GEMM builds in build_ab/ (VARIANTS), tools/bench_nt.py, plain / bias columns only
set -e
L=dino_amd/lib/libdino_hip.so
cp $L /tmp/keep.so
for v in ${VARIANTS:-new}; do
  cp build_ab/$v.so $L
  TAG=$v timeout -k 10 150 python tools/bench_nt.py | grep "M=94720" | cut -c1-110
done
cp /tmp/keep.so $L
