#!/bin/bash
# same-box timing of attention-backward builds in build_ab/ (VARIANTS), tools/bench_attn.py
set -e
L=dino_amd/lib/libdino_hip.so
cp $L /tmp/keep.so
for v in ${VARIANTS:-attn_base}; do
  cp build_ab/$v.so $L
  TAG=$v timeout -k 10 100 python tools/bench_attn.py
done
cp /tmp/keep.so $L
