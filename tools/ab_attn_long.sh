#!/bin/bash
# long-sequence attention (N = 785, ViT-S/8 B = 64): variants in build_ab/*.so against each other (kernel-level)
set -e
L=dino_amd/lib/libdino_hip.so
cp $L /tmp/keep.so
for v in ${VARIANTS:-new abl_ATTN_DMA}; do
  cp build_ab/$v.so $L
  TAG=$v timeout -k 10 100 python tools/bench_attn.py 785 128 6
done
cp /tmp/keep.so $L
