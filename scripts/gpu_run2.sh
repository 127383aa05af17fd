#!/bin/bash
# round 3, run 2: first execution of the bf16 x 3 engine (small-size oracle tests first), allocation-order probe, full GPU suite
set -x
export TMPDIR=/tmp
O=gpurun_out/r03b
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 900 python -m pytest tests/test_gpu_nets.py -m gpu -x -q -k "bf16x3" > $O/x3_small.log 2>&1; echo "rc=$?" >> $O/x3_small.log; tail -15 $O/x3_small.log
timeout 600 python scripts/alloc_order_probe.py > $O/alloc_probe.txt 2>&1; cat $O/alloc_probe.txt | grep -v amdgpu.ids
timeout 900 python scripts/layer_times.py 32 resnet50 bf16x3 > $O/layers_x3.txt 2>&1; tail -90 $O/layers_x3.txt
timeout 1800 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; tail -30 $O/gputest.log
timeout 900 python bench.py --precision bf16x3 --legs f32 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_x3.json 2> $O/bench_x3.log; tail -8 $O/bench_x3.log
