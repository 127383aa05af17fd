#!/bin/bash
tag=${1:-r06e}
o=gpurun_out/$tag
mkdir -p $o
cd "${GRAFT_REPO_ROOT:?}"
for u in 4 6 8 4 6 8; do
  SEMDEPTH_KNN_U=$u timeout 300 python bench.py --legs none --no-cpu-baseline --steps 6 --detail $o/d_$u.json > $o/b_$u.json 2> $o/b_$u.log
  echo "U=$u $(grep 'frames/s' $o/b_$u.log | cut -c1-230)"
done
SEMDEPTH_KNN_U=6 timeout 600 python -m pytest tests/test_gpu_pcl.py -q -m gpu -x 2>&1 | tail -n 2
