#!/bin/bash
# usage (GPU box, repo root): bash tools/profile_other_configs.sh r03 "vits8 vitl14 vitb16"
# BASELINE.json configs[2..4]: kernel-trace stats + the separate counter passes (FETCH_SIZE, WRITE_SIZE, MFMA busy)
# over tools/bench_other_configs.py; summaries land in gpurun_out/<tag>_<config>_*.csv for profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; cfgs=${2:-"vits8 vitl14"}
cd /tmp && export TMPDIR=/tmp && cd $R
for c in $cfgs; do
  out=$R/gpurun_out/${tag}_$c; rm -rf $out; mkdir -p $out
  DINO_TEACHER_STREAM=0 DINO_WG_STREAM=0 timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o b -- python3 $R/tools/bench_other_configs.py $c > $out/trace.log 2>&1 || { echo "trace failed $c"; exit 1; }
  cp $out/trace/b_kernel_stats.csv $R/gpurun_out/${tag}_${c}_kernel_stats.csv
  for p in FETCH_SIZE WRITE_SIZE; do
    DINO_TEACHER_STREAM=0 DINO_WG_STREAM=0 timeout -s KILL 240 rocprofv3 --pmc $p --output-format csv -d $out/$p -- python3 $R/tools/bench_other_configs.py $c > $out/$p.log 2>&1 || { echo "pass failed: $c $p"; exit 1; }
  done
  python3 $R/tools/hbm_traffic.py $out/FETCH_SIZE $out/WRITE_SIZE 8 > $R/gpurun_out/${tag}_${c}_hbm_traffic_per_kernel.csv 2> $out/traffic_summary.txt
  DINO_TEACHER_STREAM=0 DINO_WG_STREAM=0 timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $out/mfma -- python3 $R/tools/bench_other_configs.py $c > $out/mfma.log 2>&1 || { echo "pass failed: $c mfma"; exit 1; }
  python3 $R/tools/mfma_busy.py $out/mfma > $R/gpurun_out/${tag}_${c}_mfma_busy_per_kernel.csv
  timeout -s KILL 240 python3 $R/tools/bench_other_configs.py $c 2>/dev/null | tail -1 >> $R/gpurun_out/${tag}_other_configs.log
  tail -1 $R/gpurun_out/${tag}_other_configs.log
done
