#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (mean over dispatches)."""
import csv, sys, collections
path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(acc.items()):
    if not name.startswith("qed::"):
        continue
    print(name)
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
