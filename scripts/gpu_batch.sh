#!/bin/bash
tag=${1:-r04z}
mkdir -p gpurun_out/$tag
o=gpurun_out/$tag
SD_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 4 --warmup 1 --repeats 2 --no-cpu-baseline > $o/bench_2ranks.json 2> $o/bench_2ranks.log; echo "rc=$?"; grep 'frames/s' $o/bench_2ranks.log | cut -c1-200; tail -c 600 $o/bench_2ranks.json | head -c 600; echo
SD_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --config 5 --from-disk --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline > $o/bench_2ranks_disk.json 2> $o/bench_2ranks_disk.log; echo "rc=$?"; grep 'frames/s' $o/bench_2ranks_disk.log | cut -c1-200
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 3 --warmup 1 --repeats 1 --legs none --no-cpu-baseline > $o/bench_torchrun1.json 2> $o/bench_torchrun1.log; echo "rc=$?"; grep 'frames/s' $o/bench_torchrun1.log | cut -c1-160
for b in 2 8 32; do timeout 300 python scripts/layer_times.py $b resnet50 bf16x3 2> $o/layers_b$b.txt >/dev/null; echo "B=$b: $(grep -h 'dec/iconv2\|dec/upconv2\|dec/tail1\|conv1_2 \|conv2_2 ' $o/layers_b$b.txt | awk '{print $2, $6}' | tr '\n' ' ')"; done
