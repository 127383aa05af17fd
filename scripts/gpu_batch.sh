#!/bin/bash
tag=${1:-r04t}
mkdir -p gpurun_out/$tag
o=gpurun_out/$tag
timeout 1200 python -m pytest tests/test_gpu_nets.py tests/test_gpu_geometries.py -x -q -m gpu > $o/pytest.txt 2>&1; tail -4 $o/pytest.txt
timeout 300 python scripts/layer_times.py 32 resnet50 bf16x3 2> $o/layers.txt >/dev/null
SEMDEPTH_NO_FLAT=1 timeout 300 python scripts/layer_times.py 32 resnet50 bf16x3 2> $o/layers_noflat.txt >/dev/null
paste <(grep "conv_dma3" $o/layers.txt | awk '{printf "%-24s %8s\n", $2, $6}') <(grep "conv_dma3" $o/layers_noflat.txt | awk '{print $6}') | grep "fc6\|fc7\|upconv\|res4_6/conv2\|res4_2"
tail -2 $o/layers.txt; tail -2 $o/layers_noflat.txt
