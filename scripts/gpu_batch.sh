#!/bin/bash
tag=${1:-r04o}
mkdir -p gpurun_out/$tag
o=gpurun_out/$tag
for rep in 1 2; do
for v in "" "SEMDEPTH_NO_FOLD=1" "SEMDEPTH_NO_ROWSKIP=1" "SEMDEPTH_NO_TAIL1=1"; do
  env $v timeout 600 python bench.py --legs none --steps 12 --warmup 4 --repeats 2 --no-cpu-baseline > $o/bench_${v}_$rep.json 2> $o/bench_${v}_$rep.log
  echo "[$v] $(grep 'frames/s' $o/bench_${v}_$rep.log | cut -c1-200)"
done
done
