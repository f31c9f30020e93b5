#!/usr/bin/env python3
"""Per-kernel means of the counters in a rocprofv3 --pmc results database (rocpd sqlite), x the number of counter
instances per dispatch (rocprofv3 stores one row per instance: the totals are sum over instances).

    python scripts/pmc_db_summary.py gpurun_out/.../pmc_results.db [kernel-name-filter]
"""
import collections
import sqlite3
import sys

path = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else "qed"
db = sqlite3.connect(path)
cur = db.cursor()
tabs = {r[0].split("_0000")[0]: r[0] for r in cur.execute("select name from sqlite_master where type='table'")}
pe, ip, kd, ks = (tabs[k] for k in ("rocpd_pmc_event", "rocpd_info_pmc", "rocpd_kernel_dispatch", "rocpd_info_kernel_symbol"))
q = f"""select s.kernel_name, p.name, sum(e.value), count(distinct d.id) from {pe} e join {ip} p on e.pmc_id = p.id
        join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id group by s.kernel_name, p.name"""
acc = collections.defaultdict(dict)
for k, n, v, c in cur.execute(q):
    acc[k][n] = (v / max(c, 1), c)
for k in sorted(acc):
    if filt not in k:
        continue
    print(k.split("(")[0][:90])
    for n, (v, c) in sorted(acc[k].items()):
        print(f"    {n:26s} {v:14.5g}   per dispatch, {c} dispatches")
