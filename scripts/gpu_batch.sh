#!/bin/bash
# GPU batch of the moment (rewritten per experiment; results land under gpurun_out/<tag>/)
tag=${1:-r04a}
mkdir -p gpurun_out/$tag
o=gpurun_out/$tag
./scripts/probe_mfma_planes > $o/mfma_planes.txt 2>&1; cat $o/mfma_planes.txt
for d in 0 1 2; do
  SEMDEPTH_X3_DIAG=$d timeout 300 python scripts/layer_times.py 32 resnet50 bf16x3 2> $o/layers_diag$d.txt >/dev/null
  echo "diag $d: $(tail -2 $o/layers_diag$d.txt | tr '\n' ' ')"
done
for ov in "" "--no-overlap"; do
  timeout 600 python bench.py --legs none --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline $ov > $o/bench$ov.json 2> $o/bench$ov.log
  echo "[$ov] $(grep 'frames/s' $o/bench$ov.log | cut -c1-200)"
done
