#!/bin/bash
# round 3, run 13: conv_dma3 four-pair plane ring (three phases of look-ahead): parity of the bf16x3 nets, same-box A/B against the previous library, layer times
set -x
export TMPDIR=/tmp
O=gpurun_out/r03m
mkdir -p $O
L=semantic_depth_amd
timeout 1500 python -m pytest tests/test_gpu_nets.py tests/test_gpu_pipeline.py -m gpu -q -x -k "bf16x3 or x3" > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; tail -4 $O/gputest.log
cp $L/libsemdepth.so /tmp/new.so; cp $L/libsemdepth_prev.so /tmp/prev.so
for i in 1 2; do
  for v in prev new; do
    cp /tmp/$v.so $L/libsemdepth.so
    SEMDEPTH_SKIP_HASH_CHECK=1 timeout 600 python bench.py --precision bf16x3 --legs none --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.log; grep "frames/s" $O/bench_${v}_$i.log | cut -c1-200
  done
done
for v in prev new; do
  cp /tmp/$v.so $L/libsemdepth.so
  SEMDEPTH_SKIP_HASH_CHECK=1 timeout 600 python scripts/layer_times.py 32 resnet50 bf16x3 > $O/layers_$v.txt 2>&1; tail -2 $O/layers_$v.txt
done
cp /tmp/new.so $L/libsemdepth.so
