#!/bin/bash
tag=${1:-r06k}
o=gpurun_out/$tag
mkdir -p $o
cd "${GRAFT_REPO_ROOT:?}"
timeout 900 python -m pytest tests/test_gpu_nets.py -q -m gpu -x -k "two_workgroups or 16x16x32 or does_not_depend" > $o/pytest_gemm2.txt 2>&1; tail -n 8 $o/pytest_gemm2.txt
for i in 1 2; do
  SEMDEPTH_DISABLE=gemm2 timeout 300 python scripts/layer_times.py 32 resnet50 f16x2 2> $o/layer_times_ring_$i.txt >/dev/null; tail -n 2 $o/layer_times_ring_$i.txt
  timeout 300 python scripts/layer_times.py 32 resnet50 f16x2 2> $o/layer_times_gemm2_$i.txt >/dev/null; tail -n 2 $o/layer_times_gemm2_$i.txt
done
SEMDEPTH_DISABLE=gemm2 timeout 300 python bench.py --legs none --no-cpu-baseline --steps 10 --detail $o/detail_ring.json > $o/bench_ring.json 2> $o/bench_ring.log; grep 'frames/s' $o/bench_ring.log | cut -c1-200
timeout 300 python bench.py --legs none --no-cpu-baseline --steps 10 --detail $o/detail_gemm2.json > $o/bench_gemm2.json 2> $o/bench_gemm2.log; grep 'frames/s' $o/bench_gemm2.log | cut -c1-200
