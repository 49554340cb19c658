#!/bin/bash
# second SQ pass for the attention kernels: MFMA busy, LDS bank conflicts, wait classes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && cd $R
out=$R/gpurun_out/pmc_attn; mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/p2 -- python3 $R/tools/run_attn.py ${ATTN_N:-197} ${ATTN_BC:-512} > $out/p2.log 2>&1 || { echo "pass failed"; tail -5 $out/p2.log; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/p2/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_BUSY_CYCLES": n[k] += 1
for k, v in acc.items():
    if "attn" not in k: continue
    print(k, "launches", n[k])
    for c, x in sorted(v.items()): print("   %-26s %.3e per launch" % (c, x / max(n[k], 1)))
PY
