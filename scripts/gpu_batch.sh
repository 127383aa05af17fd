#!/bin/bash
# Round-5 GPU batch (rewritten per experiment; results under gpurun_out/<tag>/).  The evidence batch of the round (rocprofv3 stats, PMC passes, layer times,
# float64 check, default bench line) is this file at commit 4d78de0 ("profiles + DESIGN: evidence of the final binary").
tag=${1:-r05z}
o=gpurun_out/$tag
mkdir -p $o
timeout 3000 python -m pytest tests -q -m gpu > $o/pytest_gpu.txt 2>&1; tail -n 4 $o/pytest_gpu.txt
timeout 1200 python bench.py > $o/bench_default.json 2> $o/bench_default.log; grep 'frames/s' $o/bench_default.log | cut -c1-200
timeout 300 python scripts/layer_times.py 32 resnet50 bf16x3 2> $o/layer_times_bf16x3.txt > /dev/null; tail -n 2 $o/layer_times_bf16x3.txt
python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.txt 2>&1; tail -n 1 $o/smoke.txt
