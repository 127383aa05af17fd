#!/bin/bash
tag=${1:-r06g}
o=gpurun_out/$tag
mkdir -p $o
cd "${GRAFT_REPO_ROOT:?}"
timeout 1200 python -m pytest tests/test_gpu_nets.py tests/test_gpu_pipeline.py tests/test_gpu_geometries.py -q -m gpu -x -k "f16x2 or three_product or generic_kernels" > $o/pytest_f16x2.txt 2>&1; tail -n 6 $o/pytest_f16x2.txt
timeout 300 python scripts/layer_times.py 32 resnet50 f16x2 2> $o/layer_times_f16x2.txt >/dev/null; grep "upconv\|conv ms" $o/layer_times_f16x2.txt | cut -c1-150
SEMDEPTH_DISABLE=fold timeout 300 python scripts/layer_times.py 32 resnet50 f16x2 2> $o/layer_times_f16x2_nofold.txt >/dev/null; grep "upconv\|conv ms" $o/layer_times_f16x2_nofold.txt | cut -c1-150
timeout 300 python bench.py --legs none --no-cpu-baseline --steps 10 --detail $o/detail_f16x2.json > $o/bench_f16x2.json 2> $o/bench_f16x2.log
grep 'frames/s' $o/bench_f16x2.log | cut -c1-220
