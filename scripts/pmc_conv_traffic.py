#!/usr/bin/env python3
"""HBM traffic per conv launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same bench command.
usage: pmc_conv_traffic.py fetch_counter_collection.csv write_counter_collection.csv out.json [kernel-substring [source note]]
(kernel-substring, default "conv_": which kernels to aggregate -- e.g. "fuse_onepass" for the fusion stage)
FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM
section); WRITE_SIZE is left as reported."""
import collections, csv, json, sys

def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        if FILTER not in k or "smalln" in k:
            continue
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"]) * 1024.0
    return agg

FILTER = sys.argv[4] if len(sys.argv) > 4 else "conv_"
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
import re


def label(k):
    """the bench's profiling label (conv_*_kernel_name in csrc) of a demangled instantiation"""
    m = re.search(r"sd::(conv_\w+)_kernel<([^>]*)>", k)
    if not m:
        m2 = re.search(r"sd::(\w+_kernel)", k)
        return m2.group(1) if m2 else None
    fam, args = m.group(1), [a.strip() for a in m.group(2).split(",")]
    if fam == "conv_direct3":      # <NB, UP, KEEP, WSLOTS, FOLD>  (bf16 x 3)
        if len(args) > 4 and args[4] == "true":
            return f"conv_direct_x3_fold_kernel<{args[0]}>"
        return f"conv_direct_x3_kernel<{args[0]},2>"
    if fam == "conv_direct":       # <NB, MT, F16, N16, UP, W1, X2, H2>
        if len(args) > 7 and args[7] == "true":          # the three-product HS form (SD_PREC_F16X2)
            return "conv_direct_hs_kernel" + ("<1,n16>" if args[3] == "true" else ("<1,2>" if args[0] == "1" else "<2,2>"))
        prec = "_f16w_x2" if (len(args) > 6 and args[6] == "true") else "_f16x1" if (len(args) > 5 and args[5] == "true") else ("_f16w" if args[2] == "true" else "")
        shape = "<1,n16>" if args[3] == "true" else ("<1,2>" if args[0] == "1" else "<2,2>")
        return f"conv_direct{prec}_kernel{shape}"
    if fam == "conv_dma":          # <WM, WN, MT, NT, SIMPLE, STAGES, F16, W1, X3, H2>
        if len(args) > 9 and args[9] == "true":
            return f"conv_dma_hs_kernel<{args[0]},{args[1]},{args[2]},{args[3]}>"
        if len(args) > 8 and args[8] == "true":
            return f"conv_dma_x3_kernel<{args[0]},{args[1]},{args[2]},{args[3]}>"
        prec = "_f16x1" if (len(args) > 7 and args[7] == "true") else ("_f16w" if args[6] == "true" else "")
        return f"conv_dma{prec}_kernel<{args[0]},{args[1]},{args[2]},{args[3]}>"
    if fam == "conv_stem":         # <NB, RW, F16, X3, H2>
        if len(args) > 4 and args[4] == "true":
            return "conv_stem_hs_kernel"
        if len(args) > 3 and args[3] == "true":
            return "conv_stem_x3_kernel"
        return "conv_stem_f16w_kernel" if args[2] == "true" else "conv_stem_kernel"
    if fam == "conv_igemm":        # <WM, WN, MT, NT, VEC>
        return "conv_igemm_kernel<" + ",".join(args) + ">"
    return fam + "_kernel"


by_label = collections.defaultdict(lambda: [0, 0.0, 0.0])
for k, v in per.items():
    lb = label(k)
    if lb:
        by_label[lb][0] += v["launches"]
        by_label[lb][1] += v["fetch_bytes_per_launch"] * v["launches"]
        by_label[lb][2] += v["write_bytes_per_launch"] * v["launches"]
out = {
    "by_label": {lb: {"launches": n, "fetch_bytes_per_launch": fb / n, "write_bytes_per_launch": wb / n, "hbm_bytes_per_launch": (fb + wb) / n}
                 for lb, (n, fb, wb) in by_label.items()},
    "source": sys.argv[5] if len(sys.argv) > 5 else "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 1 --repeats 1 --no-cpu-baseline --legs none",
    "units": "FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B: MI355X_MICROARCH.md HBM section); WRITE_SIZE uncorrected",
    "per_kernel": per,
    "all_conv": {"launches": tl, "hbm_bytes_per_launch": (tf + tw) / max(tl, 1), "fetch_bytes_per_launch": tf / max(tl, 1),
                 "write_bytes_per_launch": tw / max(tl, 1)},
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["all_conv"]))
