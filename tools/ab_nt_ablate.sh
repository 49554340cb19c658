#!/bin/bash
# Timing-only ablations of gemm_nt_bf16_v3 (results are garbage): build_ab/abl_DMA.so has the loader waves
# issue no DMA, build_ab/abl_MFMA.so has the MFMA waves issue no MFMA; built with -DNT3_ABLATE_DMA / _MFMA.
set -e
L=dino_amd/lib/libdino_hip.so
cp $L /tmp/keep.so
for v in new abl_DMA abl_MFMA; do
  cp build_ab/$v.so $L
  TAG=$v timeout -k 10 150 python tools/bench_nt.py | grep -v colsum
done
cp /tmp/keep.so $L
