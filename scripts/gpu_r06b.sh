#!/bin/bash
# round 6: the register epilogues (conv_direct REGEP, conv_dma3 ep_hs, conv_dma H2): parity tests, layer times, bench of the f16x2 engine
tag=${1:-r06b}
o=gpurun_out/$tag
mkdir -p $o
cd "${GRAFT_REPO_ROOT:?}"
timeout 1500 python -m pytest tests/test_gpu_nets.py tests/test_gpu_pipeline.py tests/test_gpu_geometries.py -q -m gpu -x --durations=8 -k "f16x2 or three_product or weight_scale or fp16_range" > $o/pytest_f16x2.txt 2>&1; tail -n 6 $o/pytest_f16x2.txt
timeout 300 python scripts/layer_times.py 32 resnet50 f16x2 2> $o/layer_times_f16x2.txt >/dev/null; tail -n 3 $o/layer_times_f16x2.txt
timeout 300 python bench.py --legs none --no-cpu-baseline --steps 10 --detail $o/detail_f16x2.json > $o/bench_f16x2.json 2> $o/bench_f16x2.log
grep 'frames/s' $o/bench_f16x2.log | cut -c1-220
