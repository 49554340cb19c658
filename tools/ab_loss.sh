#!/bin/bash
# A/B of the fused loss kernel's forms in one call: one workgroup per image on/off x non-temporal row loads on/off
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for one in 0 1; do for nt in 0 1; do
  echo "DINO_LOSS_ONE=$one DINO_LOSS_NT=$nt"
  for rep in 1 2; do DINO_LOSS_ONE=$one DINO_LOSS_NT=$nt timeout -k 10 120 python tools/bench_loss.py 2>&1 | grep "us per call" || exit 1; done
done; done
