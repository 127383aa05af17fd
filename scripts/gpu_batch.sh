#!/bin/bash
# Round-5 GPU batch (rewritten per experiment; results under gpurun_out/<tag>/).
tag=${1:-r05i}
o=gpurun_out/$tag
mkdir -p $o
for rep in 1 2; do
  for v in 0 1; do
    if [ $v = 1 ]; then export SEMDEPTH_PRIO_YOUNG=1; else unset SEMDEPTH_PRIO_YOUNG; fi
    timeout 900 python bench.py --legs f16x2 --no-cpu-baseline --no-overlap > $o/bench_prio${v}_$rep.json 2> $o/bench_prio${v}_$rep.log; echo "prio_young=$v rep $rep"; grep 'frames/s' $o/bench_prio${v}_$rep.log | cut -c1-120
  done
done
