#!/bin/bash
tag=${1:-r04y}
mkdir -p gpurun_out/$tag
o=gpurun_out/$tag
for r in 1 2; do
timeout 300 python scripts/layer_times.py 32 resnet50 bf16x3 2> $o/layers_m$r.txt >/dev/null; tail -2 $o/layers_m$r.txt | tr '\n' ' '; echo
SEMDEPTH_DMA3_NFAST=1 timeout 300 python scripts/layer_times.py 32 resnet50 bf16x3 2> $o/layers_n$r.txt >/dev/null; tail -2 $o/layers_n$r.txt | tr '\n' ' '; echo
done
paste <(grep "conv_dma3" $o/layers_m1.txt | awk '{printf "%-24s %8s\n", $2, $6}') <(grep "conv_dma3" $o/layers_n1.txt | awk '{print $6}')
