#!/bin/bash
# usage: tools/pmc_gemm.sh <tag> <args to prof_one.py...>   (run from repo root on the GPU box)
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVES" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" ; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_$tag/$n -- python3 $R/tools/prof_one.py "$@" > $R/gpurun_out/pmc_$tag/$n.log 2>&1 || { echo "pass failed: $set"; exit 1; }
done
echo done
