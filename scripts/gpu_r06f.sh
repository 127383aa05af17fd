#!/bin/bash
tag=${1:-r06f}
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/$tag/pytest_gpu.txt 2>&1; tail -n 14 gpurun_out/$tag/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$tag/smoke.txt 2>&1; tail -n 1 gpurun_out/$tag/smoke.txt
./scripts/gpu_batch.sh $tag
