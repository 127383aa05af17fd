#!/bin/bash
# round 3, run 33: conv_dma cross-barrier prefetch also for the three-stage blocks: plan parity + same-box A/B
set -x
export TMPDIR=/tmp
O=gpurun_out/r03ad
mkdir -p $O
L=semantic_depth_amd
timeout 1500 python -m pytest tests/test_gpu_nets.py tests/test_gpu_geometries.py -m gpu -q -x > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; tail -3 $O/gputest.log
cp $L/libsemdepth.so /tmp/new.so; cp $L/libsemdepth_prev.so /tmp/prev.so
for i in 1 2; do
  for v in prev new; do
    cp /tmp/$v.so $L/libsemdepth.so
    SEMDEPTH_SKIP_HASH_CHECK=1 timeout 600 python bench.py --precision plan --legs none --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline > $O/bench_plan_${v}_$i.json 2> $O/bench_plan_${v}_$i.log; echo "$v $i: $(grep 'frames/s' $O/bench_plan_${v}_$i.log | cut -c1-150)"
  done
done
cp /tmp/new.so $L/libsemdepth.so
