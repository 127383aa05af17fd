#!/bin/bash
tag=${1:-r05r}
o=gpurun_out/$tag
mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "weights_in_registers" > $o/pytest_wreg.txt 2>&1; tail -n 6 $o/pytest_wreg.txt
