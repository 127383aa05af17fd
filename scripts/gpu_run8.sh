#!/bin/bash
# round 3, run 8: decomposition of conv_direct3 (no stores / no MFMAs) on the full-resolution layers and the dominant ones
set -x
export TMPDIR=/tmp
O=gpurun_out/r03h
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" || exit 1
for d in 0 1 2 3; do
  SEMDEPTH_X3_DIAG=$d timeout 600 python scripts/layer_times.py 32 resnet50 bf16x3 > $O/layers_diag$d.txt 2>&1
  echo "== diag $d"; grep -E "conv1_2 |conv3_2 |conv4_2 |upconv2|iconv2|upconv1|iconv1|res3_1/conv2" $O/layers_diag$d.txt | awk '{print $2, $6, $7}' | tr '\n' ';'; echo
done
