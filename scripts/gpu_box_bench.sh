#!/bin/bash
# the headline of the tree on whatever box the call lands on (profiles/r05_bench_final_binary_by_box.txt): gpurun --timeout 900 -- './scripts/gpu_box_bench.sh <tag>'
o=gpurun_out/${1:-r05box}
mkdir -p $o
timeout 600 python bench.py --legs none --no-cpu-baseline > $o/bench.json 2> $o/bench.log; grep 'frames/s' $o/bench.log | cut -c1-160
if [ "$2" = "smoke" ]; then timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.txt 2>&1; tail -n 1 $o/smoke.txt; fi
