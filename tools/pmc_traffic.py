#!/usr/bin/env python3
"""Combine two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; CSV output) into HBM bytes per launch of the igemm
kernels.  gfx950: FETCH_SIZE reports half the bytes of wide streaming reads -> doubled (MI355X_MICROARCH.md, HBM).
    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> backbone batch h w dtype"""
import csv, json, sys
def load(path, counter):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if any(k in r["Kernel_Name"] for k in ("igemm_kernel", "pw_kernel", "hconv_kernel", "pair_kernel", "pairw_kernel", "pairx_kernel", "pairs_kernel", "c3_kernel", "c3w_kernel", "stem_kernel")) and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"]); n += 1
    return tot, n
f, nf = load(sys.argv[1], "FETCH_SIZE")
w, nw = load(sys.argv[2], "WRITE_SIZE")
assert nf == nw and nf > 0, (nf, nw)
per_launch = (2.0 * f + w) * 1024.0 / nf
out = {"workload": [sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), sys.argv[8]],
       "igemm_launches_profiled": nf, "fetch_kb_raw_sum": f, "write_kb_sum": w,
       "igemm_hbm_bytes_per_launch": per_launch,
       "note": "HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB averaged over all igemm launches of the profiled steps"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(out)
