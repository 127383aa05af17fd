#!/bin/bash
# round 3, run 37: --overlap (tail of step i on a side stream under the convolutions of step i+1) on the headline engine, same box
mkdir -p gpurun_out/r03ah
for i in 1 2; do
  for o in "" "--overlap"; do
    timeout 600 python bench.py --precision bf16x3 --legs none --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline $o > gpurun_out/r03ah/bench${o}_$i.json 2> gpurun_out/r03ah/bench${o}_$i.log; echo "[$o] $(grep 'frames/s' gpurun_out/r03ah/bench${o}_$i.log | cut -c1-170)"
  done
done
