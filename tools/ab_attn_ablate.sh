#!/bin/bash
# Timing-only ablations of attn_bwd_dq_kernel's key step (results are garbage): no exp2 / no dQ MFMAs + transposed reads
set -e
L=dino_amd/lib/libdino_hip.so
cp $L /tmp/keep.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-attn_base attn_EXP attn_DQMMA}; do
  cp build_ab/$v.so $L
  for pw in 4 6; do
    DINO_ATTN_BWD_PW=$pw timeout -s KILL 100 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/attn_abl/$v$pw -o t -- python3 tools/run_attn.py 197 512 > /dev/null 2>&1
    echo "$v pw$pw: $(grep bwd_dq gpurun_out/attn_abl/$v$pw/t_kernel_stats.csv | awk -F, '{print "dq avg ns", $4}')  $(grep bwd_dkv gpurun_out/attn_abl/$v$pw/t_kernel_stats.csv | awk -F, '{print "dkv avg ns", $4}')"
  done
done
cp /tmp/keep.so $L
