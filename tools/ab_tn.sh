#!/bin/bash
# grouped weight-gradient launch: shipped library against timing-only builds (no DMA / no MFMA)
set -e
L=dino_amd/lib/libdino_hip.so
cp $L /tmp/keep.so
for v in ${VARIANTS:-new abl_TN_DMA abl_TN_MFMA}; do
  cp build_ab/$v.so $L
  TAG=$v timeout -k 10 100 python tools/bench_tn_group.py
done
cp /tmp/keep.so $L
