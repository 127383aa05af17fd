#!/bin/bash
# Round-5 GPU batch (rewritten per experiment; results under gpurun_out/<tag>/).
tag=${1:-r05h}
o=gpurun_out/$tag
mkdir -p $o
timeout 2400 python -m pytest tests -q -m gpu -x > $o/pytest_gpu.txt 2>&1; tail -n 6 $o/pytest_gpu.txt
timeout 900 python bench.py --legs f16x2,plan --no-cpu-baseline > $o/bench_legs.json 2> $o/bench_legs.log; grep 'frames/s' $o/bench_legs.log | cut -c1-220
timeout 300 python bench.py --precision bf16x2 --legs none --no-cpu-baseline > $o/bench_bf16x2.json 2> $o/bench_bf16x2.log; grep 'frames/s' $o/bench_bf16x2.log | cut -c1-220
