#!/bin/bash
tag=${1:-r04x}
mkdir -p gpurun_out/$tag
o=gpurun_out/$tag
timeout 1500 python -m pytest tests -x -q -m gpu > $o/pytest.txt 2>&1; tail -3 $o/pytest.txt
timeout 1200 python bench.py > $o/bench_default.json 2> $o/bench_default.log
grep 'frames/s' $o/bench_default.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
