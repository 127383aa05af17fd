#!/bin/bash
o=gpurun_out/${1:-r05box}
mkdir -p $o
timeout 600 python bench.py --legs none --no-cpu-baseline > $o/bench.json 2> $o/bench.log; grep 'frames/s' $o/bench.log | cut -c1-160
