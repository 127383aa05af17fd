#!/bin/bash
# round 3, run 15: evidence of the round-3 binary: full GPU suite, default bench line, rocprofv3 stats + PMC of the headline engine
set -x
export TMPDIR=/tmp
O=gpurun_out/r03w
mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; tail -4 $O/gputest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.log; tail -12 $O/bench_default.log
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_x3 -o x3 --output-format csv -- python3 bench.py --precision bf16x3 --legs none --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline > $O/prof_x3.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_x3_fetch -o f --output-format csv -- python3 bench.py --precision bf16x3 --legs none --steps 1 --warmup 1 --repeats 1 --no-cpu-baseline > $O/pmc_x3_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_x3_write -o w --output-format csv -- python3 bench.py --precision bf16x3 --legs none --steps 1 --warmup 1 --repeats 1 --no-cpu-baseline > $O/pmc_x3_write.log 2>&1
python scripts/pmc_conv_traffic.py $O/pmc_x3_fetch/f_counter_collection.csv $O/pmc_x3_write/w_counter_collection.csv $O/r03_pmc_conv_traffic_bf16x3.json conv_ "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --precision bf16x3 --legs none --steps 1 --warmup 1 --repeats 1 --no-cpu-baseline"
cp $O/prof_x3/x3_kernel_stats.csv $O/r03_bench_b32_bf16x3_kernel_stats.csv; head -8 $O/r03_bench_b32_bf16x3_kernel_stats.csv | cut -c1-180
timeout 600 python scripts/layer_times.py 32 resnet50 bf16x3 > $O/layers_x3.txt 2>&1; tail -2 $O/layers_x3.txt
timeout 600 python bench.py --precision bf16x3 --legs none --encoder vgg --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_x3_vgg.json 2> $O/bench_x3_vgg.log; tail -2 $O/bench_x3_vgg.log
rm -f $O/prof_x3/*trace.csv
