#!/usr/bin/env python3
"""HBM traffic per conv launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same bench command.
usage: pmc_conv_traffic.py fetch_counter_collection.csv write_counter_collection.csv out.json
FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM
section); WRITE_SIZE is left as reported."""
import collections, csv, json, sys

def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        if "conv_" not in k or "smalln" in k:
            continue
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"]) * 1024.0
    return agg

f = load(sys.argv[1], "FETCH_SIZE")
w = load(sys.argv[2], "WRITE_SIZE")
per = {}
tl = tf = tw = 0
for k in f:
    n = f[k][0]
    fb = 2.0 * f[k][1]
    wb = w.get(k, [0, 0.0])[1]
    per[k] = {"launches": n, "fetch_bytes_per_launch": fb / n, "write_bytes_per_launch": wb / n}
    tl += n; tf += fb; tw += wb
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline (bf16x2)",
    "units": "FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B: MI355X_MICROARCH.md HBM section); WRITE_SIZE uncorrected",
    "per_kernel": per,
    "all_conv": {"launches": tl, "hbm_bytes_per_launch": (tf + tw) / max(tl, 1), "fetch_bytes_per_launch": tf / max(tl, 1),
                 "write_bytes_per_launch": tw / max(tl, 1)},
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["all_conv"]))
