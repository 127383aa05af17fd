#!/bin/bash
tag=${1:-r05y}
o=gpurun_out/$tag
mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "sub_planar or fcn8s_matches_oracle" > $o/pytest.txt 2>&1; tail -n 4 $o/pytest.txt
timeout 300 python scripts/layer_times.py 32 resnet50 bf16x3 2> $o/lt_planar.txt > /dev/null; grep "conv1_1\|conv1_2\|conv2_1\|fcn conv" $o/lt_planar.txt | cut -c1-130
SEMDEPTH_NO_PLANAR=1 timeout 300 python scripts/layer_times.py 32 resnet50 bf16x3 2> $o/lt_noplanar.txt > /dev/null; grep "conv1_1\|conv1_2\|conv2_1\|fcn conv" $o/lt_noplanar.txt | cut -c1-130
