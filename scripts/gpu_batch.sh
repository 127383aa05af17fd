#!/bin/bash
# Round-5 GPU batch (rewritten per experiment; results under gpurun_out/<tag>/).
tag=${1:-r05b}
o=gpurun_out/$tag
mkdir -p $o
# 1. clean decomposition of the bf16x3 conv kernels (inputs undisturbed)
timeout 600 python scripts/decompose_x3.py 32 2> $o/decompose_x3.txt > /dev/null
grep -c sd_profile $o/decompose_x3.txt
# 2. the tests of this round's first changes
timeout 1200 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "precomputed or generic_kernels or folded_upconvs or row_grouped" > $o/pytest_nets.txt 2>&1; tail -n 5 $o/pytest_nets.txt
