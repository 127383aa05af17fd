#!/bin/bash
# round 3, run 3: evidence for the bf16 x 3 headline: default bench line, kernel stats, PMC traffic, fp32-grade check vs float64, feed rate
set -x
export TMPDIR=/tmp
O=gpurun_out/r03c
mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python -m pytest tests/test_gpu_nets.py -m gpu -x -q -k "bf16x3" > $O/x3_small.log 2>&1; tail -3 $O/x3_small.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.log; tail -12 $O/bench_default.log
timeout 900 python scripts/f32_grade_check.py > $O/f32_grade_check.txt 2>&1; cat $O/f32_grade_check.txt | grep -v amdgpu
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_x3 -o x3 --output-format csv -- python3 bench.py --precision bf16x3 --legs none --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline > $O/prof_x3.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_x3_fetch -o f --output-format csv -- python3 bench.py --precision bf16x3 --legs none --steps 1 --warmup 1 --repeats 1 --no-cpu-baseline > $O/pmc_x3_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_x3_write -o w --output-format csv -- python3 bench.py --precision bf16x3 --legs none --steps 1 --warmup 1 --repeats 1 --no-cpu-baseline > $O/pmc_x3_write.log 2>&1
python scripts/pmc_conv_traffic.py $O/pmc_x3_fetch/f_counter_collection.csv $O/pmc_x3_write/w_counter_collection.csv $O/r03_pmc_conv_traffic_bf16x3.json conv_ "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --precision bf16x3 --legs none --steps 1 --warmup 1 --repeats 1 --no-cpu-baseline"
cp $O/prof_x3/x3_kernel_stats.csv $O/r03_bench_b32_bf16x3_kernel_stats.csv; head -14 $O/r03_bench_b32_bf16x3_kernel_stats.csv | cut -c1-200
timeout 900 python scripts/layer_times.py 32 resnet50 bf16x3 > $O/layers_x3.txt 2>&1; tail -3 $O/layers_x3.txt
timeout 900 python scripts/feed_rate.py --out $O/r03_feed_rate.json > $O/feed.log 2>&1; tail -16 $O/feed.log
timeout 900 python scripts/plan_error_sweep.py 6 --precision bf16x3 > $O/strict_bf16x3.txt 2>&1; tail -2 $O/strict_bf16x3.txt
rm -f $O/prof_x3/*trace.csv
du -sh $O
