#!/bin/bash
tag=${1:-r05n}
o=gpurun_out/$tag
mkdir -p $o
timeout 300 python scripts/dma3_timed.py bf16x3 > $o/dma3_timed_bf16x3.txt 2>&1
timeout 300 python scripts/dma3_timed.py f16x2 > $o/dma3_timed_f16x2.txt 2>&1
grep -c timed $o/dma3_timed_bf16x3.txt $o/dma3_timed_f16x2.txt
