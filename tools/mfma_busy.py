#!/usr/bin/env python3
"""Per-kernel MFMA utilisation from one rocprofv3 counter pass over bench.py:
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INSTS_VALU \\
            SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d DIR -- python3 bench.py ...
Usage: mfma_busy.py DIR > out.csv"""
import csv, glob, os, sys
from collections import defaultdict

f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1]
tot, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(int)
for r in csv.DictReader(open(f)):
    tot[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[r["Kernel_Name"]] += 1
out = []
for k, c in tot.items():
    if c["GRBM_GUI_ACTIVE"] <= 0:
        continue
    # SQ_VALU_MFMA_BUSY_CYCLES sums over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE sums over the 8 XCDs
    # (cross-check: the grouped TN kernel reads 0.51 here and runs at 1 230 of 2 500 TFLOP/s)
    util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)
    out.append((c["GRBM_GUI_ACTIVE"], k, cnt[k], util, c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(c["SQ_INSTS_MFMA"], 1),
                c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_INSTS_LDS"], 1)))
w = csv.writer(sys.stdout)
w.writerow(["kernel", "launches", "MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)",
            "busy cycles per MFMA instruction", "LDS bank-conflict cycles per LDS instruction",
            "share of GPU-active cycles"])
T = sum(o[0] for o in out)
for o in sorted(out, reverse=True)[:30]:
    w.writerow([o[1], o[2], round(o[3], 3), round(o[4], 2), round(o[5], 3), round(o[0] / T, 3)])
