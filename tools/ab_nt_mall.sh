#!/bin/bash
set -e
L=dino_amd/lib/libdino_hip.so
cp $L /tmp/keep.so
for v in new abl_MFMA; do
  cp build_ab/$v.so $L
  TAG=$v PYTHONPATH=tools timeout -k 10 150 python tools/bench_nt_mall.py
done
cp /tmp/keep.so $L
