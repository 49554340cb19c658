#!/bin/bash
# usage: VARIANTS="new r33 ..." tools/ab_variants.sh <bench script.py>   (GPU box, repo root)
# Runs one bench tool against several builds of the library (build_ab/<name>.so); restores the shipped one.
set -e
L=dino_amd/lib/libdino_hip.so
cp $L /tmp/keep.so
trap "cp /tmp/keep.so $L" EXIT
for v in ${VARIANTS:-new}; do
  cp build_ab/$v.so $L
  TAG=$v timeout -k 10 ${TMO:-150} python $1
done
