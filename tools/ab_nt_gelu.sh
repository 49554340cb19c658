#!/bin/bash
# GELU-side NT GEMM shapes at the headline token count: the shipped library against timing-only builds
# (build_ab/abl_*.so; -DNT3_ABLATE_STORE drops the output stores).  Restores the shipped library afterwards.
set -e
L=dino_amd/lib/libdino_hip.so
cp $L /tmp/keep.so
for v in ${VARIANTS:-new abl_STORE}; do
  cp build_ab/$v.so $L
  TAG=$v timeout -k 10 150 python tools/bench_nt_gelu.py
done
cp /tmp/keep.so $L
