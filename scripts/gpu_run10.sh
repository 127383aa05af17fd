#!/bin/bash
# round 3, run 10: LDS reads software-pipelined ahead of the MFMAs in conv_direct3 / conv_dma3: parity of the bf16x3 engine, layer times, bench
set -x
export TMPDIR=/tmp
O=gpurun_out/r03j
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_nets.py tests/test_gpu_pipeline.py -m gpu -q -k "bf16x3 or x3" > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; tail -4 $O/gputest.log
timeout 600 python scripts/layer_times.py 32 resnet50 bf16x3 > $O/layers_x3.txt 2>&1; tail -2 $O/layers_x3.txt
timeout 600 python bench.py --precision bf16x3 --legs none --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_x3.json 2> $O/bench_x3.log; tail -3 $O/bench_x3.log
SEMDEPTH_X3_KEEP=0 timeout 600 python bench.py --precision bf16x3 --legs none --steps 10 --warmup 5 --repeats 1 --no-cpu-baseline > $O/bench_x3_nokeep.json 2> $O/bench_x3_nokeep.log; tail -3 $O/bench_x3_nokeep.log
