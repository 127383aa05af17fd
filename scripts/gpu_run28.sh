#!/bin/bash
# round 3, run 28: conv_direct (plan / bf16x2 engines) with the LDS reads software-pipelined: parity of the reduced-precision nets, same-box A/B
set -x
export TMPDIR=/tmp
O=gpurun_out/r03aa
mkdir -p $O
L=semantic_depth_amd
timeout 1500 python -m pytest tests/test_gpu_nets.py tests/test_gpu_pipeline.py tests/test_gpu_geometries.py -m gpu -q -x -k "plan or bf16x2 or mixed or geometr" > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; tail -3 $O/gputest.log
cp $L/libsemdepth.so /tmp/new.so; cp $L/libsemdepth_prev.so /tmp/prev.so
for i in 1 2; do
  for v in prev new; do
    cp /tmp/$v.so $L/libsemdepth.so
    for pr in plan bf16x2; do
      SEMDEPTH_SKIP_HASH_CHECK=1 timeout 600 python bench.py --precision $pr --legs none --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline > $O/bench_${pr}_${v}_$i.json 2> $O/bench_${pr}_${v}_$i.log; echo "$v $i: $(grep 'frames/s' $O/bench_${pr}_${v}_$i.log | cut -c1-150)"
    done
  done
done
for v in prev new; do
  cp /tmp/$v.so $L/libsemdepth.so
  SEMDEPTH_SKIP_HASH_CHECK=1 timeout 600 python scripts/layer_times.py 32 resnet50 plan > $O/layers_$v.txt 2>&1; tail -2 $O/layers_$v.txt
done
cp /tmp/new.so $L/libsemdepth.so
