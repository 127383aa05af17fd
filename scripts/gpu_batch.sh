#!/bin/bash
# Round-5 GPU batch (rewritten per experiment; results under gpurun_out/<tag>/).  The evidence batch of the round (rocprofv3 stats, PMC passes, layer times,
# float64 check, default bench line) is this file at commit 4d78de0 ("profiles + DESIGN: evidence of the final binary").
tag=${1:-r05x}
o=gpurun_out/$tag
mkdir -p $o
timeout 1200 python bench.py > $o/bench_default.json 2> $o/bench_default.log; grep 'frames/s' $o/bench_default.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.txt 2>&1; tail -n 2 $o/smoke.txt
