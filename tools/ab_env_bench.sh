#!/bin/bash
# usage: tools/ab_env_bench.sh "VAR=a" "VAR=b" ...   (GPU box, repo root): the step under different environment settings
# of ONE build, alternating, ROUNDS times each (boxes differ by ~2 %: compare within one call)
for round in $(seq 1 ${ROUNDS:-2}); do
  for e in "$@"; do
    env $e timeout -k 10 120 python bench.py --no-cpu-baseline --no-roofline --no-h2d-leg --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', d['ms_per_step'], d['value'])"
  done
done
