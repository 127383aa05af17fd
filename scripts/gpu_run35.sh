#!/bin/bash
# round 3, run 35: smoke + the -m gpu suite as the driver runs it
mkdir -p gpurun_out/r03af
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r03af/smoke.log 2>&1; tail -1 gpurun_out/r03af/smoke.log
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r03af/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03af/gputest.log; tail -4 gpurun_out/r03af/gputest.log
