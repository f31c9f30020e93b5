#!/usr/bin/env python3
"""From a rocprofv3 kernel_trace.csv: per-step busy time, idle gaps and the largest gaps between consecutive
kernels of the steady-state (graph replay) region."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last_n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
# find the last `last_n` occurrences of composite_bwd = step boundaries
idx = [i for i, r in enumerate(rows) if "composite_bwd" in r["Kernel_Name"]]
idx = idx[-last_n - 1:]
seg = rows[idx[0]:idx[-1]]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(rows[idx[-1]]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
n = len(idx) - 1
print(f"steps {n}: wall {(t1 - t0) / n / 1e3:.1f} us/step, kernel busy {busy / n / 1e3:.1f} us/step, idle {(t1 - t0 - busy) / n / 1e3:.1f} us/step, kernels/step {len(seg) / n:.1f}")
gaps = defaultdict(list)
for a, b in zip(seg[:-1], seg[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    key = (a["Kernel_Name"].split("(")[0][-40:], b["Kernel_Name"].split("(")[0][-40:])
    gaps[key].append(g)
tot = sorted(((sum(v) / n / 1e3, len(v) / n, k) for k, v in gaps.items()), reverse=True)
for t, c, k in tot[:14]:
    print(f"  {t:7.2f} us/step  x{c:4.1f}  {k[0]} -> {k[1]}")
dur = defaultdict(list)
for r in seg:
    dur[r["Kernel_Name"].split("(")[0][-48:]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for t, c, k in sorted(((sum(v) / n / 1e3, len(v) / n, k) for k, v in dur.items()), reverse=True)[:24]:
    print(f"  {t:7.2f} us/step  x{c:4.1f}  {k}")
