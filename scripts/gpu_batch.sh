#!/bin/bash
# Round-5 GPU batch (rewritten per experiment; results under gpurun_out/<tag>/).  The evidence batch of the round (rocprofv3 stats, PMC passes, layer times,
# float64 check, default bench line) is this file at commit 4d78de0 ("profiles + DESIGN: evidence of the final binary").
tag=${1:-r05k}
o=gpurun_out/$tag
mkdir -p $o
timeout 3000 python -m pytest tests -q -m gpu --durations=12 -x > $o/pytest_gpu.txt 2>&1; tail -n 22 $o/pytest_gpu.txt
timeout 1200 python bench.py > $o/bench_default.json 2> $o/bench_default.log; grep 'frames/s' $o/bench_default.log | cut -c1-200
SEMDEPTH_MFMA32=1 timeout 600 python bench.py --legs none --no-cpu-baseline > $o/bench_m32.json 2> $o/bench_m32.log; grep 'frames/s' $o/bench_m32.log | cut -c1-200
timeout 600 python bench.py --legs none --no-cpu-baseline > $o/bench_s16.json 2> $o/bench_s16.log; grep 'frames/s' $o/bench_s16.log | cut -c1-200
