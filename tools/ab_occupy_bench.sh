#!/bin/bash
# The step with and without CUs held during backward (bench.py --occupy), static tile shares against the work queue
# (GPU box, repo root).  usage: tools/ab_occupy_bench.sh [N,US[,EVERY]]
occ=${1:-16,600}
for q in 0 1; do
  for o in "" "--occupy $occ"; do
    DINO_NT_QUEUE=$q timeout -k 10 120 python bench.py --no-cpu-baseline --no-roofline --no-h2d-leg --steps 30 $o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('work queue', d['gemm_work_queue'], '| occupied', d.get('occupied_cus'), '|', d['ms_per_step'], 'ms/step', d['value'], 'img/s')"
  done
done
