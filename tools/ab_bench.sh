#!/bin/bash
# Same-box A/B of two builds of libdino_hip.so: build_ab/prev.so vs build_ab/new.so (boxes differ by
# ~2 %, and a box sometimes runs a whole bench 25 % slow: compare the per-build minima over ROUNDS).
set -e
L=dino_amd/lib/libdino_hip.so
for round in $(seq 1 ${ROUNDS:-2}); do
  for v in prev new; do
    cp build_ab/$v.so $L
    timeout -k 10 120 python bench.py --no-cpu-baseline --no-roofline --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['value'])"
    rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|junction" | tr '\n' ' ' | cut -c1-300; echo
  done
done
cp build_ab/new.so $L
