#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, per counter, mean over dispatches."""
import csv, sys, collections
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("#", path)
    for k, cs in acc.items():
        for c, v in cs.items():
            print(f"{k:60s} {c:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
