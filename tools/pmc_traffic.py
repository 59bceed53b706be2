#!/usr/bin/env python3
"""Combine two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; CSV output, mangled kernel names: -M) into HBM bytes per launch BY DEVICE
SYMBOL.  gfx950: FETCH_SIZE reports half the bytes of wide streaming reads -> doubled (MI355X_MICROARCH.md, HBM).  bench.py looks the
dominant kernels up in `by_symbol` (the symbol is what hipKernelNameRefByPtr / rocprofv3 -M print).
    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> backbone batch h w dtype"""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ursonet_amd.build import source_hash
IGEMM = ("igemm_kernel", "pw_kernel", "pwx_kernel", "hconv_kernel", "hconv2_kernel", "pair_kernel", "pairw_kernel", "pairx_kernel", "pairs_kernel", "c3_kernel", "c3w_kernel", "c3v_kernel", "stem_kernel", "stem_pool_kernel", "bneck_fwd_kernel", "bneck_dgrad_kernel", "dense_kernel", "dense_multi_kernel")


def load(path, counter):
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            k = r["Kernel_Name"][:-3] if r["Kernel_Name"].endswith(".kd") else r["Kernel_Name"]     # rocprofv3 -M prints the kernel DESCRIPTOR symbol (<name>.kd)
            tot[k] += float(r["Counter_Value"]); n[k] += 1
    return tot, n


f, nf = load(sys.argv[1], "FETCH_SIZE")
w, nw = load(sys.argv[2], "WRITE_SIZE")
by = {}
for k in f:
    if k in w and nf[k] == nw[k] and nf[k] > 0:
        by[k] = {"launches_profiled": nf[k], "fetch_kb_raw_sum": f[k], "write_kb_sum": w[k],
                 "hbm_bytes_per_launch": (2.0 * f[k] + w[k]) * 1024.0 / nf[k],
                 "fetch_bytes_per_launch": 2.0 * f[k] * 1024.0 / nf[k], "write_bytes_per_launch": w[k] * 1024.0 / nf[k]}
ig = [k for k in by if any(s in k for s in IGEMM)]
nig = sum(by[k]["launches_profiled"] for k in ig)
out = {"workload": [sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), sys.argv[8]],
       # SHA-256 of the kernel sources + flags + compiler the counters were collected on (ursonet_amd/build.py): bench.py reports these figures only
       # for the library they were measured with
       "source_hash": source_hash(),
       "note": "HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB per launch, separate --pmc passes, kernels keyed by mangled device symbol",
       "igemm_launches_profiled": nig,
       "igemm_hbm_bytes_per_launch": sum(by[k]["hbm_bytes_per_launch"] * by[k]["launches_profiled"] for k in ig) / max(nig, 1),
       "wgrad_partial_write_bytes_per_step_note": "sum over wgrad_tr*/pairw/pairx/c3g/stemw WRITE; reduce_partials_batch_kernel FETCH = partial bytes read back",
       "by_symbol": dict(sorted(by.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_profiled"]))}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in list(out["by_symbol"].items())[:12]:
    print("%-90s n=%4d  %8.1f MB/launch (fetch %8.1f, write %8.1f)" % (k[:90], v["launches_profiled"], v["hbm_bytes_per_launch"] / 1e6,
                                                                       v["fetch_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6))
