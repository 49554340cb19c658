#!/bin/bash
# Runs bench.py a few times with a background sampler of the GPU's clock and power (read-only
# rocm-smi queries): tells a throttled box from a slow build.
out=${1:-gpurun_out/clock_watch.log}
: > $out
( while true; do echo "$(date +%s.%N | cut -c1-14) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|mclk|Average Graphics Package Power|Current Socket' | sed 's/.*: //' | tr '\n' ' ')" >> $out; sleep 0.5; done ) &
SAMPLER=$!
trap "kill $SAMPLER" EXIT
for i in $(seq 1 ${RUNS:-4}); do
  echo "run $i start $(date +%s.%N | cut -c1-14)" >> $out
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-roofline --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])" >> $out
  echo "run $i end $(date +%s.%N | cut -c1-14)" >> $out
done
