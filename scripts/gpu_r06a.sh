#!/bin/bash
# round 6, first GPU call: the hardening tests of f16x2, the 8-pair float64 gate, the default bench line (does it parse, how long is it), the reserved-CU experiment
tag=${1:-r06a}
o=gpurun_out/$tag
mkdir -p $o
cd "${GRAFT_REPO_ROOT:?}"
timeout 900 python -m pytest tests/test_gpu_nets.py -q -m gpu -x --durations=8 -k "per_layer_weight_scale or weight_scale_is_shared or beyond_the_fp16_range or three_product or folded_upconvs_of_the_three or fp32_grade" > $o/pytest_new.txt 2>&1; tail -n 15 $o/pytest_new.txt
timeout 900 python scripts/f32_grade_check.py --pairs 8 --extra > $o/f32_grade_check.txt 2> $o/f32_grade_check.log; tail -n 8 $o/f32_grade_check.txt
for r in 0 8 16 4; do
  timeout 300 python bench.py --legs none --no-cpu-baseline --steps 10 --reserve-cus $r --detail $o/detail_reserve_$r.json > $o/bench_reserve_$r.json 2> $o/bench_reserve_$r.log
  grep 'frames/s' $o/bench_reserve_$r.log | cut -c1-220
done
timeout 1200 python bench.py --steps 20 --detail $o/bench_detail.json > $o/bench_default.json 2> $o/bench_default.log
grep 'frames/s\|bytes' $o/bench_default.log | cut -c1-220
wc -c $o/bench_default.json
