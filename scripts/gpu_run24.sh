#!/bin/bash
# round 3, run 24: bf16x3 heads with the f32 halo tile reconstructed once per chunk: parity of the bf16x3 nets, kernel times
set -x
export TMPDIR=/tmp
O=gpurun_out/r03x
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_nets.py tests/test_gpu_pipeline.py -m gpu -q -x -k "bf16x3 or x3" > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; tail -3 $O/gputest.log
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python3 bench.py --precision bf16x3 --legs none --steps 3 --warmup 2 --repeats 1 --no-cpu-baseline > $O/prof.log 2>&1
grep -E "frames/s" $O/prof.log | cut -c1-200
grep -E "smalln|maxpool|conv_stem" $O/prof/p_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
rm -f $O/prof/p_kernel_trace.csv
