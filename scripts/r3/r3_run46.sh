#!/bin/bash
# threshold for the block path on small graphs: 1536 / 2048 / 2708 nodes, full default epochs
mkdir -p gpurun_out
L=gpurun_out/r3_small_quality4.log; : > $L
for spec in "1536 3 15360 30" "2048 3 20480 30" "2708 2 27080 30" "4096 4 40960 30"; do
  set -- $spec
  echo "== BA $1 x $2, $3 walks x $4 epochs" >> $L
  timeout 900 python scripts/quality_probe.py --nodes $1 --m $2 --walks $3 --epochs $4 --round-walks $3 \
     --modes atomic,blocks:1:8::st 2>&1 | grep -v amdgpu.ids >> $L
done
cat $L
