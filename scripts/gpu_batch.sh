#!/bin/bash
tag=${1:-r04w}
mkdir -p gpurun_out/$tag
o=gpurun_out/$tag
timeout 900 python -m pytest tests/test_gpu_nets.py tests/test_gpu_geometries.py -x -q -m gpu > $o/pytest.txt 2>&1; tail -3 $o/pytest.txt
for r in 1 2; do
timeout 300 python scripts/layer_times.py 32 resnet50 bf16x3 2> $o/layers_new$r.txt >/dev/null; grep -h "conv1_1 \|enc/conv1 " $o/layers_new$r.txt | awk '{print $2, $6}' | tr '\n' ' '; tail -2 $o/layers_new$r.txt | tr '\n' ' '; echo
done
cp scripts/ab/conv_stem_old.hip semantic_depth_amd/csrc/conv_stem.hip
python -m semantic_depth_amd.build > $o/build_old.log 2>&1
for r in 1 2; do
timeout 300 python scripts/layer_times.py 32 resnet50 bf16x3 2> $o/layers_old$r.txt >/dev/null; grep -h "conv1_1 \|enc/conv1 " $o/layers_old$r.txt | awk '{print $2, $6}' | tr '\n' ' '; tail -2 $o/layers_old$r.txt | tr '\n' ' '; echo
done
