#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) counter-collection run: mean counter value per kernel name.
usage: python tools/pmc_summary.py <results.db> [kernel-substring]"""
import sqlite3, sys, collections, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
filt = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
rows = cur.execute("select * from counters_collection").fetchall()
ix = {c: i for i, c in enumerate(cols)}
kn = ix.get("kernel_name", ix.get("name")); cn = ix["counter_name"]; cv = ix.get("value", ix.get("counter_value"))
grid = ix.get("grid_size")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if filt in r[kn]:
        key = (r[kn], r[grid] if grid is not None else 0)
        agg[key][r[cn]].append(r[cv])
for (k, g), d in agg.items():
    short = re.sub(r"\(.*", "", k)
    m = re.search(r"gemm_i8_\w+|quant_\w+", k); abl = re.search(r"Li(\d+)EE", k)
    print(f"{m.group(0) if m else short[:60]} ABL={abl.group(1) if abl else '-'} grid={g}")
    for c, v in sorted(d.items()):
        print(f"    {c:34s} mean {sum(v)/len(v):16.1f}  (n={len(v)})")
