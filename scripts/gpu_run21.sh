#!/bin/bash
# dev: dump the raw road cloud of bench frame 0
mkdir -p gpurun_out/r03u
timeout 600 python bench.py --precision plan --legs none --no-cpu-baseline --dump-cloud gpurun_out/r03u/road0.npy 2>&1 | tail -3
