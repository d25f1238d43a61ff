#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection CSVs under a directory: per (kernel, counter) mean over dispatches."""
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    if "gemm" in k or "attn" in k:
        print(f"{k:60s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4e}")
