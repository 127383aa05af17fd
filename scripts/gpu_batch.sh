#!/bin/bash
# GPU batch of the moment (rewritten per experiment; results land under gpurun_out/<tag>/)
tag=${1:-r04c}
mkdir -p gpurun_out/$tag
o=gpurun_out/$tag
timeout 900 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -s -k "folded or fp32_grade or full_size_split" > $o/pytest.txt 2>&1; tail -25 $o/pytest.txt
timeout 300 python scripts/layer_times.py 32 resnet50 bf16x3 2> $o/layers.txt >/dev/null
grep -h "dec/\|conv ms" $o/layers.txt
timeout 600 python bench.py --legs none --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline > $o/bench.json 2> $o/bench.log
echo "$(grep 'frames/s' $o/bench.log | cut -c1-200)"
