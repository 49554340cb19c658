#!/bin/bash
# GPU box: 192 x 384 eight-wave weight-gradient kernel (DINO_TN_V3=1) against the shipped 128 x 384 one: parity, then timing
set -e
DINO_TN_V3=1 timeout -k 10 400 python -u -m pytest tests/test_gpu_ops.py -x -q -k "gemm_tn or gemm_group" --timeout 120 --timeout-method thread 2>&1 | tail -4
for r in 1 2; do
  DINO_TN_V3=0 TAG=tn_v2 timeout -k 10 100 python tools/bench_tn_group.py
  DINO_TN_V3=1 TAG=tn_v3 timeout -k 10 100 python tools/bench_tn_group.py
done
