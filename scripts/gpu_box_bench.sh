#!/bin/bash
# Same-box regression harness: the headline engine and the legs of TWO trees, back to back on whatever box the call lands on.
#     gpurun --timeout 1500 -- './scripts/gpu_box_bench.sh <tag> [<other tree, default .r05tree>]'
# The other tree is an export of an earlier commit with its own library built in place (git archive <commit> | tar -x -C .r05tree; python -m semantic_depth_amd.build there):
# it travels with the snapshot (.gitignore'd, not .gpurunignore'd).  A, B, A order: the first tree runs twice, so drift over the call shows.
o=$PWD/gpurun_out/${1:-r06box}
other=${2:-.r05tree}
mkdir -p $o
run() {  # tree label
  ( cd $1 && timeout 600 python bench.py --precision f16x2 --legs bf16x3,plan --no-cpu-baseline --steps 10 > $o/bench_$2.json 2> $o/bench_$2.log )
  grep 'frames/s' $o/bench_$2.log | cut -c1-150 | sed "s/^/$2 /"
}
run . this_1
[ -d "$other" ] && run $other other
run . this_2
python - "$o" <<'PY'
import re, sys, glob, os
o = sys.argv[1]
rows = {}
for f in sorted(glob.glob(os.path.join(o, "bench_*.log"))):
    lab = os.path.basename(f)[6:-4]
    for m in re.finditer(r"\[(\w+)\] ([\d.]+) frames/s \(([\d.]+) ms/step", open(f).read()):
        rows.setdefault(m.group(1), {})[lab] = (float(m.group(2)), float(m.group(3)))
with open(os.path.join(o, "same_box_table.txt"), "w") as out:
    labs = sorted({l for r in rows.values() for l in r})
    out.write("engine      " + "".join(f"{l:>22s}" for l in labs) + "\n")
    for eng, r in rows.items():
        out.write(f"{eng:12s}" + "".join(f"{r[l][0]:12.1f} f/s {r[l][1]:6.2f}" if l in r else " " * 22 for l in labs) + "\n")
print(open(os.path.join(o, "same_box_table.txt")).read())
PY
