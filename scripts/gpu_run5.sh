#!/bin/bash
# round 3, run 5: the 256 x 256 phased GEMM block of the bf16 x 3 engine (conv_dma3): correctness at B = 32, A/B, bench
set -x
export TMPDIR=/tmp
O=gpurun_out/r03e
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 900 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -k "b32 and bf16x3" > $O/x3_b32.log 2>&1; tail -4 $O/x3_b32.log
SEMDEPTH_NO_DMA3=1 timeout 600 python scripts/layer_times.py 32 resnet50 bf16x3 > $O/layers_nodma3.txt 2>&1; tail -2 $O/layers_nodma3.txt
timeout 600 python scripts/layer_times.py 32 resnet50 bf16x3 > $O/layers_dma3.txt 2>&1; tail -2 $O/layers_dma3.txt
grep -E "fc6|fc7|res4_2/conv3|res2_2/conv3|res5_2/conv3|res4_2/conv1" $O/layers_nodma3.txt $O/layers_dma3.txt | cut -c1-200
timeout 900 python bench.py --steps 10 --warmup 3 --legs none --no-cpu-baseline > $O/bench_x3.json 2> $O/bench_x3.log; tail -3 $O/bench_x3.log
timeout 2400 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; tail -6 $O/gputest.log
