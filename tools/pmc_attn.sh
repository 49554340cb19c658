#!/bin/bash
# SQ counters of the attention kernels alone (one pass, <= 8 SQ counters): where the cycles of a wave go
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && cd $R
out=$R/gpurun_out/pmc_attn; mkdir -p $out
export DINO_ATTN_PERSIST=${DINO_ATTN_PERSIST:-0}
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_MFMA --output-format csv -d $out/p1 -- python3 $R/tools/run_attn.py ${ATTN_N:-197} ${ATTN_BC:-512} > $out/p1.log 2>&1 || { echo "pass failed"; tail -5 $out/p1.log; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/p1/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k, v in acc.items():
    if "attn" not in k: continue
    print(k, "launches", n[k])
    for c, x in sorted(v.items()): print("   %-22s %.3e per launch" % (c, x / max(n[k], 1)))
PY
