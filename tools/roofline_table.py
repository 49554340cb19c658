#!/usr/bin/env python3
"""Per-kernel roofline of one profiled bench run, from the three committed summaries of tools/profile_round.sh:
  roofline_table.py <kernel_stats.csv> <hbm_traffic_per_kernel.csv> <mfma_busy_per_kernel.csv> [steps in the trace]
For every kernel that takes > 0.3 % of the step: time per step, average launch, HBM bytes per launch from the counters,
and the two floors a launch cannot beat on this hardware --
  t_hbm  = fetched bytes / 6.3 TB/s + written bytes / 5.0 TB/s  (the streaming rates profiles/r03_hbm_ceiling.log
           measured for loads and for stores; a copy priced this way runs 5.6 TB/s, the probe's copy 5.0-5.5)
  t_mfma = MFMA-busy fraction x duration (the time the matrix pipes were actually occupied)
-- and roof = max(t_hbm, t_mfma) / duration: 1.0 means nothing but bytes or MFMAs is left in the kernel."""
import csv
import sys

RD, WR = 6.3e12, 5.0e12


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    if n.startswith("_ZN12_GLOBAL__N_1"):
        n = n[len("_ZN12_GLOBAL__N_1"):].lstrip("0123456789")
        for cut in ("ID", "IL", "Ef", "EP", "Ev"):
            i = n.find(cut)
            if i > 0:
                n = n[:i]
                break
    i = n.find("(")
    return (n[:i] if i > 0 else n)[:44]


def main():
    stats, traffic, busy = sys.argv[1:4]
    steps = float(sys.argv[4]) if len(sys.argv) > 4 else 13.0
    tr, bz = {}, {}
    with open(traffic) as f:
        rd = csv.reader(f)
        next(rd)
        for r in rd:
            if len(r) >= 6:
                tr[r[0]] = (2 * float(r[2]) * 1024, float(r[3]) * 1024)   # fetched (x2 on gfx950), written bytes
    with open(busy) as f:
        rd = csv.reader(f)
        next(rd)
        for r in rd:
            bz[r[0]] = float(r[2])
    rows, tot, tot_roof = [], 0.0, 0.0
    with open(stats) as f:
        for r in csv.DictReader(f):
            ms = float(r["TotalDurationNs"]) / steps / 1e6
            tot += ms
            us = float(r["AverageNs"]) / 1e3
            if float(r["Percentage"]) < 0.3:
                continue
            fb, wb = tr.get(r["Name"], (None, None))
            b = bz.get(r["Name"])
            t_hbm = None if fb is None else (fb / RD + wb / WR) * 1e6
            t_mfma = None if b is None else b * us
            floor = max(x for x in (t_hbm, t_mfma, 0.0) if x is not None)
            tot_roof += ms * floor / us
            rows.append((ms, short(r["Name"]), int(r["Calls"]) / steps, us, fb, wb, t_hbm, b, floor / us))
    print("| kernel | launches/step | ms/step | avg us | HBM MB read / written | TB/s | t_hbm us | MFMA busy | roof |")
    print("|---|---|---|---|---|---|---|---|---|")
    for ms, n, calls, us, fb, wb, t_hbm, b, roof in rows:
        mb = "-" if fb is None else f"{fb / 1e6:.0f} / {wb / 1e6:.0f}"
        tbs = "-" if fb is None else f"{(fb + wb) / us / 1e6:.2f}"
        print(f"| `{n}` | {calls:.0f} | {ms:.2f} | {us:.1f} | {mb} | {tbs} | "
              f"{'-' if t_hbm is None else f'{t_hbm:.0f}'} | {'-' if b is None else f'{b:.2f}'} | {roof:.2f} |")
    print(f"\nkernel time per step {tot:.2f} ms; the listed kernels at their floors {tot_roof:.2f} ms "
          f"({tot_roof / tot:.2f} of the step)")


if __name__ == "__main__":
    main()
