#!/bin/bash
tag=${1:-r04k}
mkdir -p gpurun_out/$tag
o=gpurun_out/$tag
timeout 900 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -s -k "row_grouped or fcn8s_matches or fp32_grade" > $o/pytest.txt 2>&1; tail -5 $o/pytest.txt
timeout 300 python scripts/layer_times.py 32 resnet50 bf16x3 2> $o/layers.txt >/dev/null
grep -h "fc6 \|fc7 \|conv ms" $o/layers.txt
SEMDEPTH_NO_ROWSKIP=1 timeout 300 python scripts/layer_times.py 32 resnet50 bf16x3 2> $o/layers_norowskip.txt >/dev/null
grep -h "fc6 \|conv ms" $o/layers_norowskip.txt
