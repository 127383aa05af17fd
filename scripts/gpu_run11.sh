#!/bin/bash
# round 3, run 11: same-box A/B of the software-pipelined LDS reads (libsemdepth_prev.so = the commit before) on the sustained bench
set -x
export TMPDIR=/tmp
O=gpurun_out/r03k
mkdir -p $O
L=semantic_depth_amd
cp $L/libsemdepth.so /tmp/new.so; cp $L/libsemdepth_prev.so /tmp/prev.so
for i in 1 2 3; do
  for v in prev new; do
    cp /tmp/$v.so $L/libsemdepth.so
    SEMDEPTH_SKIP_HASH_CHECK=1 timeout 600 python bench.py --precision bf16x3 --legs none --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.log; grep "frames/s" $O/bench_${v}_$i.log | cut -c1-200
  done
done
cp /tmp/new.so $L/libsemdepth.so
