#!/bin/bash
o=gpurun_out/r05soak
mkdir -p $o
timeout 900 python bench.py --legs none --no-cpu-baseline --steps 150 --warmup 5 > $o/bench_soak.json 2> $o/bench_soak.log; grep 'frames/s' $o/bench_soak.log | cut -c1-200
timeout 600 python -c "
import __graft_entry__ as g; g.smoke()" > $o/smoke.txt 2>&1; tail -n 2 $o/smoke.txt
