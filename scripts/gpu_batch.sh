#!/bin/bash
tag=${1:-r05p}
o=gpurun_out/$tag
mkdir -p $o
for rep in 1 2; do
  for v in 1 0; do
    if [ $v = 1 ]; then export SEMDEPTH_EPI_DRAIN=1; else unset SEMDEPTH_EPI_DRAIN; fi
    timeout 900 python bench.py --legs f16x2 --no-cpu-baseline --no-overlap > $o/bench_drain${v}_$rep.json 2> $o/bench_drain${v}_$rep.log; echo "epi_drain=$v rep $rep"; grep 'frames/s' $o/bench_drain${v}_$rep.log | cut -c1-120
  done
done
unset SEMDEPTH_EPI_DRAIN
timeout 1500 python -m pytest tests/test_gpu_nets.py tests/test_gpu_pipeline.py -x -q -m gpu -k "bf16x3 or f16x2 or plan or bf16x2" > $o/pytest.txt 2>&1; tail -n 4 $o/pytest.txt
