#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, mean counter value per dispatch."""
import csv, sys, collections, glob, os
paths = []
for a in sys.argv[1:]:
    paths += glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True) if os.path.isdir(a) else [a]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for pth in paths:
    for r in csv.DictReader(open(pth)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:32s} n={len(v):4d} mean={sum(v)/len(v):.4g}")
