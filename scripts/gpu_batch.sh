#!/bin/bash
# The GPU batch behind profiles/r06_*: rocprofv3 kernel stats, HBM traffic (separate FETCH / WRITE passes), SQ counters, layer times, the float64 grade check and
# the default bench line, for the headline engine (f16x2) and for the six-product leg (bf16x3), in ONE gpurun call of ~15 min:
#     gpurun --timeout 2400 -- './scripts/gpu_batch.sh r06v'      (results under gpurun_out/<tag>/)
# The test suite is its own call (one outer timeout must not have to cover both):  gpurun --timeout 1500 -- 'python -m pytest tests -q -m gpu'
tag=${1:-r06v}
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/$tag
o=gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for prec in f16x2 bf16x3; do
  B="bench.py --no-overlap --legs none --no-cpu-baseline --precision $prec"
  rocprofv3 --kernel-trace --stats -d $o/stats_$prec -o t --output-format csv -- python3 $B --steps 5 --warmup 2 --repeats 1 > $o/bench_stats_$prec.json 2> $o/bench_stats_$prec.log
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $o/fetch_$prec -o t --output-format csv -- python3 $B --steps 1 --warmup 1 --repeats 1 > /dev/null 2> $o/fetch_$prec.log
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $o/write_$prec -o t --output-format csv -- python3 $B --steps 1 --warmup 1 --repeats 1 > /dev/null 2> $o/write_$prec.log
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d $o/sq_$prec -o t --output-format csv -- python3 $B --steps 3 --warmup 2 --repeats 1 > /dev/null 2> $o/sq_$prec.log
  python3 scripts/pmc_conv_traffic.py $o/fetch_$prec/t_counter_collection.csv $o/write_$prec/t_counter_collection.csv $o/pmc_conv_traffic_$prec.json conv_ "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 $B --steps 1 --warmup 1 --repeats 1"
  python3 scripts/pmc_conv_traffic.py $o/fetch_$prec/t_counter_collection.csv $o/write_$prec/t_counter_collection.csv $o/pmc_tail_traffic_$prec.json dec_tail "same passes, dec_tail1 kernel"
  python3 scripts/pmc_conv_traffic.py $o/fetch_$prec/t_counter_collection.csv $o/write_$prec/t_counter_collection.csv $o/pmc_fuse_traffic_$prec.json fuse_onepass "same passes, fuse_onepass_kernel"
  python3 scripts/pmc_sq_summary.py $o/sq_$prec/t_counter_collection.csv $o/sq_$prec/t_kernel_trace.csv $o/pmc_sq_$prec.json "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -- python3 $B --steps 3 --warmup 2 --repeats 1" > $o/sq_summary_$prec.txt
  cp $o/stats_$prec/t_kernel_stats.csv $o/kernel_stats_$prec.csv
  rm -rf $o/fetch_$prec $o/write_$prec $o/sq_$prec $o/stats_$prec
  timeout 300 python scripts/layer_times.py 32 resnet50 $prec 2> $o/layer_times_$prec.txt >/dev/null
done
timeout 900 python scripts/f32_grade_check.py --pairs 8 --extra > $o/f32_grade_check.txt 2> $o/f32_grade_check.log
timeout 1200 python bench.py --steps 20 --detail $o/bench_detail.json > $o/bench_default.json 2> $o/bench_default.log
grep 'frames/s' $o/bench_default.log | cut -c1-200
