#!/bin/bash
# round 3, run 4: A/B of the X-fragment register cache of conv_direct3 (SEMDEPTH_X3_KEEP), host CPU quota probe, bench
set -x
export TMPDIR=/tmp
O=gpurun_out/r03d
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" || exit 1
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc; python -c "import os; print('affinity', len(os.sched_getaffinity(0)))"; grep -c processor /proc/cpuinfo; lscpu | grep -E "Model name|Thread|Core|Socket" 
for k in 0 1 2; do
  SEMDEPTH_X3_KEEP=$k timeout 600 python scripts/layer_times.py 32 resnet50 bf16x3 > $O/layers_keep$k.txt 2>&1; tail -2 $O/layers_keep$k.txt
done
timeout 600 python -m pytest tests/test_gpu_nets.py tests/test_gpu_pipeline.py -m gpu -x -q -k "bf16x3" > $O/x3_tests.log 2>&1; tail -3 $O/x3_tests.log
timeout 900 python bench.py --steps 10 --warmup 3 --legs none --no-cpu-baseline > $O/bench_x3.json 2> $O/bench_x3.log; tail -3 $O/bench_x3.log
