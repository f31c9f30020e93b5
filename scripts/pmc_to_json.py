#!/usr/bin/env python3
"""rocprofv3 --pmc result databases (rocpd sqlite) -> the JSON summaries bench.py quotes.

    python scripts/pmc_to_json.py traffic <fetch.db> <write.db> <out.json> "<kernels version / command>"
    python scripts/pmc_to_json.py valu    <valu.db>            <out.json> "<kernels version / command>"

Per kernel: totals per dispatch (sum over the counter's instances), mean over dispatches.
"""
import collections
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the sources of the kernels the counters describe: bench.py quotes a summary only while these still hash the same
KERNEL_SOURCES = ("qed_splatter_amd/csrc/composite.hip", "qed_splatter_amd/csrc/qed_common.h")


def source_hashes():
    return {rel: hashlib.sha256(open(os.path.join(ROOT, rel), "rb").read()).hexdigest() for rel in KERNEL_SOURCES}


def per_kernel(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = {r[0].split("_0000")[0]: r[0] for r in cur.execute("select name from sqlite_master where type='table'")}
    pe, ip, kd, ks = (tabs[k] for k in ("rocpd_pmc_event", "rocpd_info_pmc", "rocpd_kernel_dispatch", "rocpd_info_kernel_symbol"))
    q = f"""select s.kernel_name, p.name, sum(e.value), count(distinct d.id) from {pe} e join {ip} p on e.pmc_id = p.id
            join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id group by s.kernel_name, p.name"""
    acc = collections.defaultdict(dict)
    for k, n, v, c in cur.execute(q):
        acc[k][n] = v / max(c, 1)
    return acc


def pretty(mangled, names):
    """demangled-ish key: the names bench.py looks up"""
    for short, full in names.items():
        if short in mangled:
            return full
    return mangled.split("(")[0][:80]


NAMES = {"composite_bwd_kernelILi4": "qed::composite_bwd_kernel<4>", "composite_fwd_kernelILi4": "qed::composite_fwd_kernel<4>",
         "composite_bwd_kernelILi3": "qed::composite_bwd_kernel<3>", "composite_fwd_kernelILi3": "qed::composite_fwd_kernel<3>"}

mode = sys.argv[1]
if mode == "traffic":
    f, w, out, about = per_kernel(sys.argv[2]), per_kernel(sys.argv[3]), sys.argv[4], sys.argv[5]
    kernels = {}
    for k in f:
        if "qed" not in k:
            continue
        kernels[pretty(k, NAMES)] = {"fetch_size_kb": f[k].get("FETCH_SIZE"), "write_size_kb": w.get(k, {}).get("WRITE_SIZE")}
    json.dump({"_about": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two separate passes), totals per dispatch, KB as "
                         "reported; gfx950: FETCH_SIZE counts half the bytes of a wide coalesced read (double it), "
                         "WRITE_SIZE is exact for 16-B stores and float atomics (MI355X_MICROARCH.md)",
               "kernels_version": about, "source_sha256": source_hashes(), "kernels": kernels}, open(out, "w"), indent=1)
else:
    v, out, about = per_kernel(sys.argv[2]), sys.argv[3], sys.argv[4]
    kernels = {}
    for k in v:
        if "composite" not in k:
            continue
        c = v[k]
        simd_quads = 1024 * (c["GRBM_GUI_ACTIVE"] / 8.0) / 4.0
        kernels[pretty(k, NAMES)] = {
            "sq_active_inst_valu": c["SQ_ACTIVE_INST_VALU"], "sq_insts_valu": c["SQ_INSTS_VALU"],
            "sq_wave_cycles": c.get("SQ_WAVE_CYCLES"), "grbm_gui_active_sum_xcd": c["GRBM_GUI_ACTIVE"],
            "valu_issue_frac": round(c["SQ_ACTIVE_INST_VALU"] / simd_quads, 3),
            "mean_waves_per_simd": round(c.get("SQ_WAVE_CYCLES", 0.0) / simd_quads, 2)}
    json.dump({"_about": "valu_issue_frac = SQ_ACTIVE_INST_VALU (quad-cycles) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs / 4)",
               "kernels_version": about, "source_sha256": source_hashes(), "kernels": kernels}, open(out, "w"), indent=1)
print("wrote", out)
