#!/usr/bin/env python3
"""aggregate a rocprofv3 --pmc counter_collection csv by kernel name: usage pmc_summary.py <csv> [substr]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    k = r.get("Kernel_Name", "")
    if flt not in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    print(k[:90])
    for c, v in sorted(d.items()):
        print(f"    {c:32s} {v:.4g}  (n={cnt[(k, c)]})")
