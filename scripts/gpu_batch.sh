#!/bin/bash
tag=${1:-r04g}
mkdir -p gpurun_out/$tag
o=gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $o/prof -o trace --output-format csv -- python3 bench.py --legs none --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline > $o/bench_prof.json 2> $o/bench_prof.log
ls -R $o/prof | head -20
