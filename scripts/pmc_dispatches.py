#!/usr/bin/env python3
"""list rocprofv3 --pmc counters per dispatch for kernels matching a substring: pmc_dispatches.py <csv> <substr>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2]
d = collections.OrderedDict()
for r in rows:
    if flt not in r.get("Kernel_Name", ""):
        continue
    k = int(r["Dispatch_Id"])
    d.setdefault(k, {"grid": r.get("Grid_Size", ""), "name": r["Kernel_Name"][:60]})[r["Counter_Name"]] = float(r["Counter_Value"])
for k in sorted(d):
    v = d[k]
    print(k, v["name"], v["grid"], " ".join(f"{c}={x:.4g}" for c, x in v.items() if c not in ("grid", "name")))
