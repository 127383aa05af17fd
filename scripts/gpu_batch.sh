#!/bin/bash
tag=${1:-r04n}
mkdir -p gpurun_out/$tag
o=gpurun_out/$tag
timeout 900 python bench.py --legs none --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline > $o/bench.json 2> $o/bench.log
grep 'frames/s' $o/bench.log | cut -c1-220
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -s -k "process_frame_full_size or b8_full_size" > $o/pytest.txt 2>&1; tail -6 $o/pytest.txt
for fd in "" "--from-disk"; do
timeout 900 python bench.py --config 5 $fd --legs none --steps 10 --warmup 2 --repeats 2 --no-cpu-baseline > $o/bench5$fd.json 2> $o/bench5$fd.log
echo "config 5 $fd: $(grep 'frames/s' $o/bench5$fd.log | cut -c1-160)"; tail -2 $o/bench5$fd.log | cut -c1-300
done
