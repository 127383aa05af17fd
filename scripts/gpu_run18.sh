#!/bin/bash
# round 3, run 18: where the radius filter's time goes (diagnostic library: slow path of ror_count skipped -- wrong results, timing only)
set -x
export TMPDIR=/tmp
O=gpurun_out/r03r
mkdir -p $O
L=semantic_depth_amd
cp $L/libsemdepth.so /tmp/new.so
for v in new diag; do
  if [ $v = diag ]; then cp $L/libsemdepth_diag.so $L/libsemdepth.so; export SEMDEPTH_SKIP_HASH_CHECK=1; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$v -o p --output-format csv -- python3 bench.py --precision plan --legs none --steps 3 --warmup 2 --repeats 1 --no-cpu-baseline > $O/prof_$v.log 2>&1
  grep -E "ror_count|sor_knn|grid_|med_|plane_filter|end_points|cmp_" $O/prof_$v/p_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
  rm -f $O/prof_$v/p_kernel_trace.csv
done
cp /tmp/new.so $L/libsemdepth.so
