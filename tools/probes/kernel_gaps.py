#!/usr/bin/env python3
"""Idle time between the kernels of the replayed step: from a rocprofv3 --kernel-trace CSV (Start_Timestamp / End_Timestamp per dispatch)
take the last `--steps` replays of the training graph and report, per step, the sum of kernel durations, the sum of the gaps between
consecutive kernels and the wall time first start -> last end.    python tools/probes/kernel_gaps.py <kernel_trace.csv> [--launches 146]"""
import csv, sys, argparse
ap = argparse.ArgumentParser(); ap.add_argument("csv"); ap.add_argument("--launches", type=int, default=0); ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(a.csv))]
rows.sort()
# one step = from an `mold_kernel` launch to the next
starts = [i for i, r in enumerate(rows) if "mold_kernel" in r[2]]
steps = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)][-a.steps:]
tot_d = tot_g = tot_w = 0.0
big = {}
for s, e in steps:
    ks = rows[s:e]
    d = sum(k[1] - k[0] for k in ks)
    gaps = [(ks[i + 1][0] - ks[i][1], ks[i][2], ks[i + 1][2]) for i in range(len(ks) - 1)]
    g = sum(x[0] for x in gaps)
    w = ks[-1][1] - ks[0][0]
    tot_d += d; tot_g += g; tot_w += w
    for x in gaps:
        big.setdefault((x[1][:40], x[2][:40]), []).append(x[0])
n = len(steps)
print("steps %d  launches/step %d  kernel time %.3f ms  gaps %.3f ms  first start -> last end %.3f ms  (mean gap %.2f us)" % (
    n, len(rows[steps[0][0]:steps[0][1]]), tot_d / n / 1e6, tot_g / n / 1e6, tot_w / n / 1e6, tot_g / n / 1e3 / max(len(rows[steps[0][0]:steps[0][1]]) - 1, 1)))
top = sorted(big.items(), key=lambda kv: -sum(kv[1]) / len(kv[1]))[:15]
for (a_, b_), v in top:
    print("  gap %6.2f us  after %-40s before %s" % (sum(v) / len(v) / 1e3, a_, b_))
