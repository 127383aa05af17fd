#!/bin/bash
# Round-5 GPU batch (rewritten per experiment; results under gpurun_out/<tag>/).
tag=${1:-r05c}
o=gpurun_out/$tag
mkdir -p $o
# 1. the three-product fp16 engine against the oracle (small sizes) and against float64
timeout 1500 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "f16x2 or float64" > $o/pytest_f16x2.txt 2>&1; tail -n 15 $o/pytest_f16x2.txt
# 2. its speed: per-layer times and the bench leg
timeout 300 python scripts/layer_times.py 32 resnet50 f16x2 2> $o/layer_times_f16x2.txt > /dev/null; tail -n 2 $o/layer_times_f16x2.txt
timeout 300 python scripts/layer_times.py 32 resnet50 bf16x2 2> $o/layer_times_bf16x2.txt > /dev/null; tail -n 2 $o/layer_times_bf16x2.txt
timeout 900 python bench.py --legs f32,f16x2 --no-cpu-baseline > $o/bench_legs.json 2> $o/bench_legs.log; grep 'frames/s' $o/bench_legs.log | cut -c1-220
