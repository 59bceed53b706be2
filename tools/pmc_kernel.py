#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel from counter_collection CSVs.  python tools/pmc_kernel.py <csv>... [--match wgrad]"""
import csv, sys, collections
match = None
files = []
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == "--match": match = args.pop(0)
    else: files.append(a)
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if match and match not in r["Kernel_Name"]:
            continue
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    print(k)
    for c, (v, n) in sorted(cs.items()):
        print("   %-32s %16.0f  (avg of %d)" % (c, v / n, n))
