#!/bin/bash
# GPU box: one-pass attention backward (DINO_ATTN_ONEPASS=1) against the fused kernel: parity tests, then timing
set -e
export DINO_ATTN_ONEPASS=1
timeout -k 10 300 python -u -m pytest tests/test_gpu_ops.py -x -q -k "attention" --timeout 120 --timeout-method thread 2>&1 | tail -5
TAG=onepass timeout -k 10 100 python tools/bench_attn.py 197 512 6
export DINO_ATTN_ONEPASS=0
TAG=fused timeout -k 10 100 python tools/bench_attn.py 197 512 6
