#!/bin/bash
# usage (GPU box, repo root): bash tools/profile_round.sh r02 v1
# Kernel-trace stats + the separate counter passes MI355X_MICROARCH.md prescribes (FETCH_SIZE, WRITE_SIZE,
# MFMA busy) over the SAME short bench.py run; summaries land in gpurun_out/<tag>_*.csv for profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; ver=$2
cd /tmp && export TMPDIR=/tmp && cd $R
STEPS="--steps 3 --warmup 2 --no-cpu-baseline --no-h2d-leg --no-roofline"
out=$R/gpurun_out/${tag}_${ver}; mkdir -p $out
# per-kernel durations: one stream, so that a kernel's duration is its own
DINO_TEACHER_STREAM=0 DINO_WG_STREAM=0 timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o b -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2d-leg --no-roofline > $out/trace.log 2>&1 || { echo "trace failed"; exit 1; }
cp $out/trace/b_kernel_stats.csv $R/gpurun_out/${tag}_bench_kernel_stats_${ver}.csv
for c in FETCH_SIZE WRITE_SIZE; do
  DINO_TEACHER_STREAM=0 DINO_WG_STREAM=0 timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $out/$c -- python3 $R/bench.py $STEPS > $out/$c.log 2>&1 || { echo "pass failed: $c"; exit 1; }
done
python3 $R/tools/hbm_traffic.py $out/FETCH_SIZE $out/WRITE_SIZE 5 > $R/gpurun_out/${tag}_hbm_traffic_per_kernel_${ver}.csv 2> $out/traffic_summary.txt
cat $out/traffic_summary.txt
DINO_TEACHER_STREAM=0 DINO_WG_STREAM=0 timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $out/mfma -- python3 $R/bench.py $STEPS > $out/mfma.log 2>&1 || { echo "pass failed: mfma"; exit 1; }
python3 $R/tools/mfma_busy.py $out/mfma > $R/gpurun_out/${tag}_mfma_busy_per_kernel_${ver}.csv
# the bench line of the same build (both side streams on, as the driver runs it)
timeout -s KILL 300 python3 $R/bench.py > $R/gpurun_out/${tag}_bench_${ver}.log 2>&1
tail -1 $R/gpurun_out/${tag}_bench_${ver}.log | cut -c 1-400
