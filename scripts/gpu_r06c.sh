#!/bin/bash
# round 6: same-box A/B of the register epilogues (SEMDEPTH_LDS_EPILOGUE=1 = round 5's LDS transposition), the LDS-staged kNN: pcl parity tests + bench
tag=${1:-r06c}
o=gpurun_out/$tag
mkdir -p $o
cd "${GRAFT_REPO_ROOT:?}"
timeout 900 python -m pytest tests/test_gpu_pcl.py tests/test_gpu_fusion.py -q -m gpu -x > $o/pytest_pcl.txt 2>&1; tail -n 4 $o/pytest_pcl.txt
timeout 600 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -x -k "records or tail or process" > $o/pytest_pipe.txt 2>&1; tail -n 4 $o/pytest_pipe.txt
for i in 1 2; do
  SEMDEPTH_LDS_EPILOGUE=1 timeout 300 python scripts/layer_times.py 32 resnet50 f16x2 2> $o/layer_times_lds_$i.txt >/dev/null; tail -n 2 $o/layer_times_lds_$i.txt
  timeout 300 python scripts/layer_times.py 32 resnet50 f16x2 2> $o/layer_times_reg_$i.txt >/dev/null; tail -n 2 $o/layer_times_reg_$i.txt
done
SEMDEPTH_LDS_EPILOGUE=1 timeout 300 python bench.py --legs none --no-cpu-baseline --steps 10 --detail $o/detail_lds.json > $o/bench_lds.json 2> $o/bench_lds.log; grep 'frames/s' $o/bench_lds.log | cut -c1-220
timeout 300 python bench.py --legs none --no-cpu-baseline --steps 10 --detail $o/detail_reg.json > $o/bench_reg.json 2> $o/bench_reg.log; grep 'frames/s' $o/bench_reg.log | cut -c1-220
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $o/stats -o t --output-format csv -- python3 bench.py --no-overlap --legs none --no-cpu-baseline --steps 5 --warmup 2 --repeats 1 > $o/bench_stats.json 2> $o/bench_stats.log
cp $o/stats/t_kernel_stats.csv $o/kernel_stats_f16x2.csv; rm -rf $o/stats
grep -i "knn\|ror_count\|grid_" $o/kernel_stats_f16x2.csv | cut -c1-200
