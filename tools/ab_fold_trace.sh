#!/bin/bash
# Per-kernel durations of the step with and without the LayerNorm fold (one stream, rocprofv3 --kernel-trace --stats)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && cd $R
for f in 0 1; do
  out=$R/gpurun_out/fold_trace$f; rm -rf $out; mkdir -p $out
  DINO_LN_FOLD=$f DINO_TEACHER_STREAM=0 DINO_WG_STREAM=0 timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o b -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2d-leg --no-roofline > $out/trace.log 2>&1 || { echo "trace failed $f"; exit 1; }
  cp $out/b_kernel_stats.csv $R/gpurun_out/r03_fold${f}_kernel_stats.csv
done
