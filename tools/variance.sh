#!/bin/bash
# Run-to-run spread of bench.py under stream settings (each line: settings, then ms/step per run).
# Usage: RUNS=5 tools/variance.sh "DINO_WG_STREAM=0" "DINO_WG_DEPTH=3" ...
for cfg in "$@"; do
  line="$cfg:"
  for i in $(seq 1 ${RUNS:-5}); do
    v=$(env $cfg timeout -k 10 120 python bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])") || exit 1
    line="$line $v"
  done
  echo "$line"
done
