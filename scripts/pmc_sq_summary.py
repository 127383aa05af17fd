#!/usr/bin/env python3
"""MFMA-pipe utilisation and effective clock per kernel from one rocprofv3 pass
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- python3 bench.py ...
usage: pmc_sq_summary.py counter_collection.csv kernel_trace.csv out.json [source note]
GRBM_GUI_ACTIVE = shader-clock cycles the GPU was busy during the dispatch, reported as the SUM over the 8 XCDs (each has its own
GRBM: the raw ratio to the duration reads 14-18 "GHz"), so GRBM_GUI_ACTIVE / 8 / duration = the effective clock (MI355X_MICROARCH.md,
DVFS give-back); SQ_VALU_MFMA_BUSY_CYCLES is summed over the SIMDs' MFMA pipes (the guide: = 32 x N_mfma for 32x32x16 bf16), so
busy / (GUI_ACTIVE / 8 x 4 SIMDs x 256 CUs) = the fraction of the chip's MFMA issue capacity in use."""
XCDS = 8
import collections, csv, json, sys

cc, kt, out = sys.argv[1], sys.argv[2], sys.argv[3]
note = sys.argv[4] if len(sys.argv) > 4 else ""
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
seen = collections.defaultdict(set)
for r in csv.DictReader(open(cc)):
    k = r["Kernel_Name"]
    if "sd::conv_" not in k and "sd::dec_" not in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen[k]:
        seen[k].add(r["Dispatch_Id"])
        agg[k]["_ns"] += dur.get(r["Dispatch_Id"], (0, ""))[0]
        agg[k]["_n"] += 1
res = {}
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["_ns"]):
    ns, gui = v["_ns"], v.get("GRBM_GUI_ACTIVE", 0.0)
    e = {"launches": int(v["_n"]), "total_ms": ns / 1e6}
    if gui and ns:
        e["effective_clock_ghz"] = gui / XCDS / ns
        e["mfma_busy_frac_of_issue_capacity"] = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / XCDS * 4 * 256)
    wc = v.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in v:
                e[c.lower() + "_frac_of_wave_cycles"] = v[c] / wc
    res[k[:120]] = e
json.dump({"source": note, "kernels": res}, open(out, "w"), indent=1)
for k, e in list(res.items())[:10]:
    print(k[:70], {a: (round(b, 4) if isinstance(b, float) else b) for a, b in e.items()})
