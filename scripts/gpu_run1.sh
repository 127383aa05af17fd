#!/bin/bash
# round-3 evidence pass on one MI355X box: GPU tests, the default bench line, rocprofv3 kernel stats + PMC traffic of the f32 engine and
# of the fusion kernel, strict-error records.  Usage: gpurun --timeout 2400 -- 'bash scripts/gpu_run1.sh'
set -x
export TMPDIR=/tmp
O=gpurun_out/r03a
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -5 $O/gputest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.log
tail -12 $O/bench_default.log
# kernel stats of the f32 engine (program directly after --)
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_f32 -o f32 --output-format csv -- python3 bench.py --precision f32 --legs none --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline > $O/prof_f32.log 2>&1
# HBM traffic, separate passes
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f32_fetch -o f --output-format csv -- python3 bench.py --precision f32 --legs none --steps 1 --warmup 1 --repeats 1 --no-cpu-baseline > $O/pmc_f32_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_f32_write -o w --output-format csv -- python3 bench.py --precision f32 --legs none --steps 1 --warmup 1 --repeats 1 --no-cpu-baseline > $O/pmc_f32_write.log 2>&1
ls -R $O | head -60
F=$(find $O/pmc_f32_fetch -name '*counter_collection.csv' | head -1); Wc=$(find $O/pmc_f32_write -name '*counter_collection.csv' | head -1)
python scripts/pmc_conv_traffic.py "$F" "$Wc" $O/r03_pmc_conv_traffic_f32.json conv_ "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --precision f32 --legs none --steps 1 --warmup 1 --repeats 1 --no-cpu-baseline"
python scripts/pmc_conv_traffic.py "$F" "$Wc" $O/r03_pmc_fuse_traffic.json fuse_onepass "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --precision f32 --legs none --steps 1 --warmup 1 --repeats 1 --no-cpu-baseline (the fusion stage is the same kernel under every engine)"
S=$(find $O/prof_f32 -name '*kernel_stats.csv' | head -1); cp "$S" $O/r03_bench_b32_f32_kernel_stats.csv
head -12 $O/r03_bench_b32_f32_kernel_stats.csv
# strict per-element error records
timeout 900 python scripts/plan_error_sweep.py 10 > $O/strict_plan.txt 2>&1
timeout 600 python scripts/plan_error_sweep.py 4 --precision bf16x2 > $O/strict_bf16x2.txt 2>&1
timeout 600 python scripts/plan_error_sweep.py 6 --decoder-std 0.01 > $O/strict_plan_std001.txt 2>&1
tail -2 $O/strict_plan.txt $O/strict_bf16x2.txt $O/strict_plan_std001.txt
rm -rf $O/prof_f32/*/*.db
du -sh $O
