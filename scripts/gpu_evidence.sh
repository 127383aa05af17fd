#!/bin/bash
# The round's evidence call: the GPU test suite, smoke(), then the profile batch (scripts/gpu_batch.sh), results under gpurun_out/<tag>/:
#     gpurun --timeout 3600 -- './scripts/gpu_evidence.sh r06j'
tag=${1:-r06v}
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/$tag/pytest_gpu.txt 2>&1; tail -n 14 gpurun_out/$tag/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$tag/smoke.txt 2>&1; tail -n 1 gpurun_out/$tag/smoke.txt
./scripts/gpu_batch.sh $tag
