#!/bin/bash
# round 3, run 25: soak: 150 steps x 3 regions of the headline engine (sustained clocks, allocator stability)
set -x
export TMPDIR=/tmp
O=gpurun_out/r03y
mkdir -p $O
timeout 900 python bench.py --precision bf16x3 --legs none --steps 150 --warmup 10 --repeats 3 --no-cpu-baseline > $O/soak.json 2> $O/soak.log; grep "frames/s" $O/soak.log | cut -c1-220
python - <<'P'
import json
d=json.loads(open('gpurun_out/r03y/soak.json').read().strip().splitlines()[-1])
print(d['value'], d['repeat_ms_per_step'])
P
