#!/bin/bash
# round 3, run 6: f32 candidate prefilter of the Open3D filters (exactness tests + road-stage time), full GPU suite, default bench
set -x
export TMPDIR=/tmp
O=gpurun_out/r03f
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 900 python -m pytest tests/test_gpu_pcl.py tests/test_gpu_fusion.py -m gpu -x -q > $O/pcl.log 2>&1; tail -3 $O/pcl.log
timeout 900 python bench.py --precision plan --steps 10 --warmup 3 --legs none --no-cpu-baseline > $O/bench_plan.json 2> $O/bench_plan.log; tail -3 $O/bench_plan.log
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_plan -o plan --output-format csv -- python3 bench.py --precision plan --legs none --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline > $O/prof_plan.log 2>&1
grep -E "sor_knn|ror_count|grid_" $O/prof_plan/plan_kernel_stats.csv | cut -c1-160
timeout 2400 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; tail -4 $O/gputest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.log; tail -12 $O/bench_default.log
rm -f $O/prof_plan/*trace.csv
