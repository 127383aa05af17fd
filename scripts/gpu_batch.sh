#!/bin/bash
tag=${1:-r04m}
mkdir -p gpurun_out/$tag
o=gpurun_out/$tag
timeout 900 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -s -k "folded or fp32_grade or full_size_split or monodepth" > $o/pytest.txt 2>&1; tail -14 $o/pytest.txt
timeout 300 python scripts/layer_times.py 32 resnet50 bf16x3 2> $o/layers.txt >/dev/null
grep -h "dec/\|conv ms" $o/layers.txt
