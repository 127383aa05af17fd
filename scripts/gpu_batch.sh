#!/bin/bash
# Round-5 GPU batch (rewritten per experiment; results under gpurun_out/<tag>/).
tag=${1:-r05g}
o=gpurun_out/$tag
mkdir -p $o
timeout 2400 python -m pytest tests -q -m gpu > $o/pytest_gpu.txt 2>&1; tail -n 12 $o/pytest_gpu.txt
timeout 300 python scripts/layer_times.py 32 resnet50 f16x2 2> $o/layer_times_f16x2.txt > /dev/null; tail -n 3 $o/layer_times_f16x2.txt
timeout 900 python bench.py --legs f16x2 --no-cpu-baseline > $o/bench_legs.json 2> $o/bench_legs.log; grep 'frames/s' $o/bench_legs.log | cut -c1-220
