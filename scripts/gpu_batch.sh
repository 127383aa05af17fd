#!/bin/bash
tag=${1:-r04j}
mkdir -p gpurun_out/$tag
o=gpurun_out/$tag
timeout 1500 python -m pytest tests -x -q -m gpu > $o/pytest.txt 2>&1; tail -5 $o/pytest.txt
timeout 300 python scripts/layer_times.py 32 resnet50 bf16x3 2> $o/layers.txt >/dev/null
grep -h "dec/tail1\|conv ms" $o/layers.txt
timeout 900 python bench.py --steps 20 --warmup 5 --repeats 3 > $o/bench.json 2> $o/bench.log
grep 'frames/s' $o/bench.log | cut -c1-220
