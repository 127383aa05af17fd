#!/bin/bash
tag=${1:-r04aa}
mkdir -p gpurun_out/$tag
o=gpurun_out/$tag
timeout 900 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "precomputed or row_grouped" > $o/pytest.txt 2>&1; tail -3 $o/pytest.txt
