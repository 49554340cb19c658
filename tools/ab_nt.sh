#!/bin/bash
# Same-box A/B of the NT GEMM shapes: build_ab/prev.so vs build_ab/new.so (kernel-level, tools/bench_nt.py)
set -e
L=dino_amd/lib/libdino_hip.so
for v in prev new; do
  cp build_ab/$v.so $L
  TAG=$v timeout -k 10 150 python tools/bench_nt.py
done
cp build_ab/new.so $L
