#!/bin/bash
# Round-5 GPU batch (rewritten per experiment; results under gpurun_out/<tag>/).
tag=${1:-r05j}
o=gpurun_out/$tag
mkdir -p $o
timeout 3000 python -m pytest tests -q -m gpu > $o/pytest_gpu.txt 2>&1; tail -n 12 $o/pytest_gpu.txt
