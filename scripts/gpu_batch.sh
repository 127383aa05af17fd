#!/bin/bash
# Round-5 GPU batch (rewritten per experiment; results under gpurun_out/<tag>/).
tag=${1:-r05k}
o=gpurun_out/$tag
mkdir -p $o
timeout 600 python scripts/decompose_x3.py 32 f16x2 2> $o/decompose_f16x2.txt > /dev/null
grep -c sd_profile $o/decompose_f16x2.txt
