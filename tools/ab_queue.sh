#!/bin/bash
# static tile shares against the work queue, with and without held CUs (GPU box, repo root)
set -e
for q in 0 1; do
  DINO_NT_QUEUE=$q timeout -k 10 120 python tools/bench_nt_occupied.py
done
