#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection.csv rows per (kernel, counter) for kernels whose name contains argv[2]; prints means per dispatch.
usage: pmc_agg.py <counter_collection.csv> <kernel substring>"""
import collections, csv, sys
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        a = acc[(r["Kernel_Name"][:60], r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print(f"{k:60s} {c:28s} {v / n:16.4e}  ({n} dispatches)")
