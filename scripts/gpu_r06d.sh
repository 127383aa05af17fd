#!/bin/bash
tag=${1:-r06d}
o=gpurun_out/$tag
mkdir -p $o
cd "${GRAFT_REPO_ROOT:?}"
timeout 1500 python -m pytest tests -q -m gpu --durations=12 > $o/pytest_gpu.txt 2>&1; tail -n 22 $o/pytest_gpu.txt
./scripts/gpu_box_bench.sh $tag/box
