#!/bin/bash
# Round-5 GPU batch (rewritten per experiment; results under gpurun_out/<tag>/).  The evidence batch of the round (rocprofv3 stats, PMC passes, layer times,
# float64 check, default bench line) is this file at commit 4d78de0 ("profiles + DESIGN: evidence of the final binary").
tag=${1:-r05t}
o=gpurun_out/$tag
mkdir -p $o
timeout 600 python scripts/direct3_timed.py > $o/direct3_timed.txt 2>&1; tail -n 70 $o/direct3_timed.txt | cut -c1-400
