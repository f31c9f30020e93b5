#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly: name, calls, avg us, percentage."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("void ", "")[:64]
    if float(r["Percentage"]) > float(sys.argv[2] if len(sys.argv) > 2 else 0.3):
        print(f"{n:66s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:10.2f} {float(r['Percentage']):7.2f}")
