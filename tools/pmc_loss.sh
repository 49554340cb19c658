#!/bin/bash
# HBM traffic of the loss kernels: separate FETCH_SIZE / WRITE_SIZE passes (MI355X_MICROARCH.md), csv output
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_loss/$c -- python3 $R/tools/bench_loss.py > $R/gpurun_out/pmc_loss_$c.log 2>&1 || { echo "pass failed: $c"; exit 1; }
done
python3 $R/tools/hbm_traffic.py $R/gpurun_out/pmc_loss/FETCH_SIZE $R/gpurun_out/pmc_loss/WRITE_SIZE 23 2>/dev/null | grep -i "loss\|colsum\|kernel,"
