#!/bin/bash
tag=${1:-r06i}
o=gpurun_out/$tag
mkdir -p $o
cd "${GRAFT_REPO_ROOT:?}"
timeout 1200 python -m pytest tests/test_gpu_nets.py tests/test_gpu_pipeline.py tests/test_gpu_geometries.py -q -m gpu -x -k "f16x2 or three_product or generic_kernels or bf16x3" > $o/pytest_sel.txt 2>&1; tail -n 4 $o/pytest_sel.txt
./scripts/gpu_box_bench.sh $tag/box .prevtree
