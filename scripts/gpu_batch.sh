#!/bin/bash
tag=${1:-r04u}
mkdir -p gpurun_out/$tag
o=gpurun_out/$tag
timeout 900 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "fcn8s_matches or monodepth_matches or folded or row_grouped" > $o/pytest.txt 2>&1; tail -3 $o/pytest.txt
for r in 1 2; do
timeout 300 python scripts/layer_times.py 32 resnet50 bf16x3 2> $o/layers_new$r.txt >/dev/null; tail -2 $o/layers_new$r.txt | tr '\n' ' '; echo
done
cp semantic_depth_amd/csrc/conv_direct3.hip /tmp/new.hip
cp scripts/ab/conv_direct3_old.hip semantic_depth_amd/csrc/conv_direct3.hip
python -m semantic_depth_amd.build > $o/build_old.log 2>&1
for r in 1 2; do
timeout 300 python scripts/layer_times.py 32 resnet50 bf16x3 2> $o/layers_old$r.txt >/dev/null; tail -2 $o/layers_old$r.txt | tr '\n' ' '; echo
done
paste <(grep "conv_direct" $o/layers_new1.txt | awk '{printf "%-24s %8s\n", $2, $6}') <(grep "conv_direct" $o/layers_old1.txt | awk '{print $6}') | head -30
