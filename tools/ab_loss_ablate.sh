#!/bin/bash
# Timing-only ablations of loss_fused_bf16_kernel (results are garbage): build_ab/loss_abl_STORE.so computes the
# gradient and drops it, build_ab/loss_abl_TEACHER.so skips the teacher phase's exponentials.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
L=dino_amd/lib/libdino_hip.so
cp $L /tmp/keep.so
for v in full loss_abl_STORE loss_abl_TEACHER; do
  [ $v = full ] || cp build_ab/$v.so $L
  echo "== $v"
  timeout -k 10 120 python tools/bench_loss.py 256 65536 12 0 2>&1 | grep "us per call"
done
cp /tmp/keep.so $L
