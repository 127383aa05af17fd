#!/bin/bash
tag=${1:-r04q}
mkdir -p gpurun_out/$tag
o=gpurun_out/$tag
timeout 900 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -s -k "folded" > $o/pytest.txt 2>&1; tail -3 $o/pytest.txt
for d in 0 1 2 3; do
  SEMDEPTH_X3_DIAG=$d timeout 300 python scripts/layer_times.py 32 resnet50 bf16x3 2> $o/layers_diag$d.txt >/dev/null
  echo "diag $d: $(grep -h 'dec/tail1' $o/layers_diag$d.txt)"
done
