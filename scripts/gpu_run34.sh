#!/bin/bash
# dev: durations of the GPU suite
mkdir -p gpurun_out/r03ae
timeout 2400 python -m pytest tests -m gpu -q --durations=25 > gpurun_out/r03ae/durations.log 2>&1; tail -40 gpurun_out/r03ae/durations.log
