#!/bin/bash
# would the streamed kernels beat the whole-problem kernels at mid lengths?  (kernel-level)
for cfg in "197 512 6" "257 64 16" "145 640 6" "577 32 6"; do
  TAG=whole timeout -k 10 100 python tools/bench_attn.py $cfg || exit 1
  DINO_ATTN_STREAM_MIN=65 TAG=stream timeout -k 10 100 python tools/bench_attn.py $cfg || exit 1
done
