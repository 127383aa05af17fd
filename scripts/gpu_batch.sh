#!/bin/bash
# Round-5 GPU batch (rewritten per experiment; results under gpurun_out/<tag>/).
tag=${1:-r05e}
o=gpurun_out/$tag
mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "folded_upconvs_of_the_three" > $o/pytest_fold.txt 2>&1; tail -n 8 $o/pytest_fold.txt
timeout 300 python scripts/layer_times.py 32 resnet50 f16x2 2> $o/layer_times_f16x2.txt > /dev/null; tail -n 2 $o/layer_times_f16x2.txt
timeout 900 python bench.py --legs f16x2 --no-cpu-baseline > $o/bench_legs.json 2> $o/bench_legs.log; grep 'frames/s' $o/bench_legs.log | cut -c1-220
