#!/bin/bash
# round 3, run 7: kNN prefilter without rounding-mode switches; three-slot weight ring of the NB = 1 layers (A/B); octet pool; bench
set -x
export TMPDIR=/tmp
O=gpurun_out/r03g
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 900 python -m pytest tests/test_gpu_pcl.py tests/test_gpu_fusion.py -m gpu -x -q > $O/pcl.log 2>&1; tail -2 $O/pcl.log
timeout 900 python -m pytest tests/test_gpu_nets.py tests/test_gpu_pipeline.py -m gpu -x -q -k "bf16x3" > $O/x3.log 2>&1; tail -3 $O/x3.log
SEMDEPTH_X3_NO_RING3=1 timeout 600 python scripts/layer_times.py 32 resnet50 bf16x3 > $O/layers_ring2.txt 2>&1; tail -1 $O/layers_ring2.txt
timeout 600 python scripts/layer_times.py 32 resnet50 bf16x3 > $O/layers_ring3.txt 2>&1; tail -1 $O/layers_ring3.txt
grep -E "upconv2|iconv2|upconv1|iconv1" $O/layers_ring2.txt $O/layers_ring3.txt | cut -c1-190
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o x3 --output-format csv -- python3 bench.py --legs none --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline > $O/prof.log 2>&1
grep -E "sor_knn|ror_count|maxpool3z" $O/prof/x3_kernel_stats.csv | cut -c1-200
timeout 900 python bench.py --steps 10 --warmup 3 --legs none --no-cpu-baseline > $O/bench_x3.json 2> $O/bench_x3.log; tail -3 $O/bench_x3.log
rm -f $O/prof/*trace.csv
