#!/bin/bash
# Round-5 GPU batch (rewritten per experiment; results under gpurun_out/<tag>/).  The evidence batch of the round (rocprofv3 stats, PMC passes, layer times,
# float64 check, default bench line) is this file at commit 4d78de0 ("profiles + DESIGN: evidence of the final binary").
tag=${1:-r05z2}
o=gpurun_out/$tag
mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_nets.py -x -q -m gpu > $o/pytest_nets.txt 2>&1; tail -n 4 $o/pytest_nets.txt
SEMDEPTH_MFMA16=1 timeout 300 python scripts/dma3_timed.py bf16x3 > $o/dma3_timed_s16.txt 2>&1; grep "dma3 timed" $o/dma3_timed_s16.txt | tail -n 4 | cut -c1-250
timeout 300 python scripts/dma3_timed.py bf16x3 > $o/dma3_timed_m32.txt 2>&1; grep "dma3 timed" $o/dma3_timed_m32.txt | tail -n 4 | cut -c1-250
for rep in 1 2; do
timeout 300 python scripts/layer_times.py 32 resnet50 bf16x3 2> $o/layer_times_bf16x3_m32_$rep.txt > /dev/null; tail -n 2 $o/layer_times_bf16x3_m32_$rep.txt
SEMDEPTH_MFMA16=1 timeout 300 python scripts/layer_times.py 32 resnet50 bf16x3 2> $o/layer_times_bf16x3_s16_$rep.txt > /dev/null; tail -n 2 $o/layer_times_bf16x3_s16_$rep.txt
done
timeout 300 python scripts/layer_times.py 32 resnet50 f16x2 2> $o/layer_times_f16x2_m32.txt > /dev/null; tail -n 2 $o/layer_times_f16x2_m32.txt
SEMDEPTH_MFMA16=1 timeout 300 python scripts/layer_times.py 32 resnet50 f16x2 2> $o/layer_times_f16x2_s16.txt > /dev/null; tail -n 2 $o/layer_times_f16x2_s16.txt
