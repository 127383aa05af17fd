#!/bin/bash
# The GPU batch behind profiles/r04_*: all -m gpu tests, rocprofv3 kernel stats, HBM traffic (separate FETCH / WRITE passes), SQ counters, layer times and
# the default bench line, in ONE gpurun call:   gpurun --timeout 3600 -- './scripts/gpu_batch.sh r04v'   (results under gpurun_out/<tag>/).
# During the round this file was rewritten per experiment (A/B runs of a switch, decomposition runs with SEMDEPTH_X3_DIAG, ...).
tag=${1:-r04}
mkdir -p gpurun_out/$tag
o=gpurun_out/$tag
timeout 1500 python -m pytest tests -x -q -m gpu > $o/pytest.txt 2>&1; tail -3 $o/pytest.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="bench.py --no-overlap --legs none --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $o/stats -o t --output-format csv -- python3 $B --steps 5 --warmup 2 --repeats 1 > $o/bench_stats.json 2> $o/bench_stats.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $o/fetch -o t --output-format csv -- python3 $B --steps 1 --warmup 1 --repeats 1 > /dev/null 2> $o/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $o/write -o t --output-format csv -- python3 $B --steps 1 --warmup 1 --repeats 1 > /dev/null 2> $o/write.log
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d $o/sq -o t --output-format csv -- python3 $B --steps 3 --warmup 2 --repeats 1 > /dev/null 2> $o/sq.log
python3 scripts/pmc_conv_traffic.py $o/fetch/t_counter_collection.csv $o/write/t_counter_collection.csv $o/pmc_conv_traffic_bf16x3.json conv_ "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 $B --steps 1 --warmup 1 --repeats 1"
python3 scripts/pmc_conv_traffic.py $o/fetch/t_counter_collection.csv $o/write/t_counter_collection.csv $o/pmc_tail_traffic.json dec_tail "same passes, dec_tail1_x3_kernel"
python3 scripts/pmc_conv_traffic.py $o/fetch/t_counter_collection.csv $o/write/t_counter_collection.csv $o/pmc_fuse_traffic.json fuse_onepass "same passes, fuse_onepass_kernel"
python3 scripts/pmc_sq_summary.py $o/sq/t_counter_collection.csv $o/sq/t_kernel_trace.csv $o/pmc_sq_bf16x3.json "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -- python3 $B --steps 3 --warmup 2 --repeats 1" > $o/sq_summary.txt
cp $o/stats/t_kernel_stats.csv $o/kernel_stats.csv
rm -rf $o/fetch $o/write $o/sq $o/stats
timeout 300 python scripts/layer_times.py 32 resnet50 bf16x3 2> $o/layer_times_bf16x3.txt >/dev/null
timeout 1200 python bench.py > $o/bench_default.json 2> $o/bench_default.log
grep 'frames/s' $o/bench_default.log | cut -c1-200
