#!/bin/bash
tag=${1:-r05o}
o=gpurun_out/$tag
mkdir -p $o
timeout 300 python scripts/layer_times.py 32 resnet50 bf16x3 2> $o/layer_times_bf16x3.txt > /dev/null; tail -n 3 $o/layer_times_bf16x3.txt
timeout 900 python scripts/f32_grade_check.py > $o/f32_grade_check.txt 2> $o/f32_grade_check.log; cat $o/f32_grade_check.txt
timeout 900 python bench.py --legs f16x2 --no-cpu-baseline > $o/bench_legs.json 2> $o/bench_legs.log; grep 'frames/s' $o/bench_legs.log | cut -c1-220
timeout 3000 python -m pytest tests -q -m gpu -x > $o/pytest_gpu.txt 2>&1; tail -n 5 $o/pytest_gpu.txt
