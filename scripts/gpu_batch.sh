#!/bin/bash
tag=${1:-r05p}
o=gpurun_out/$tag
mkdir -p $o
timeout 300 python scripts/dma3_timed.py bf16x3 > $o/dma3_timed_s16.txt 2>&1; grep "dma3 timed" $o/dma3_timed_s16.txt | tail -n 6 | cut -c1-250
for rep in 1 2; do timeout 300 python scripts/layer_times.py 32 resnet50 bf16x3 2> $o/layer_times_bf16x3_$rep.txt > /dev/null; tail -n 2 $o/layer_times_bf16x3_$rep.txt; done
timeout 600 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "bf16x3" > $o/pytest_nets.txt 2>&1; tail -n 3 $o/pytest_nets.txt
