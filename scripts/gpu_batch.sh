#!/bin/bash
# Round-5 GPU batch (rewritten per experiment; results under gpurun_out/<tag>/).  The evidence batch of the round (rocprofv3 stats, PMC passes, layer times,
# float64 check, default bench line) is this file at commit 4d78de0 ("profiles + DESIGN: evidence of the final binary").
tag=${1:-r05a3}
o=gpurun_out/$tag
mkdir -p $o
timeout 600 python scripts/direct3_timed.py > $o/direct3_timed.txt 2>&1; grep "timed\]" $o/direct3_timed.txt | grep "chunks=32 items=16\|chunks=4 items=128" | tail -n 6 | cut -c1-330
timeout 300 python scripts/layer_times.py 32 resnet50 bf16x3 2> $o/layer_times_bf16x3.txt > /dev/null; tail -n 2 $o/layer_times_bf16x3.txt
timeout 600 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "bf16x3" > $o/pytest_nets.txt 2>&1; tail -n 3 $o/pytest_nets.txt
timeout 300 python scripts/layer_times.py 32 resnet50 bf16x3 2> $o/layer_times_bf16x3_2.txt > /dev/null; tail -n 2 $o/layer_times_bf16x3_2.txt
