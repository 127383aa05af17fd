#!/bin/bash
# round 3, run 20: quick loop for the Open3D filters: pcl parity tests + kernel times of the road chain
set -x
export TMPDIR=/tmp
O=gpurun_out/r03t
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pcl.py -m gpu -q -x > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; tail -3 $O/gputest.log
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python3 bench.py --precision plan --legs none --steps 3 --warmup 2 --repeats 1 --no-cpu-baseline > $O/prof.log 2>&1
grep -E "frames/s" $O/prof.log | cut -c1-200
rm -f $O/prof/p_kernel_trace.csv
