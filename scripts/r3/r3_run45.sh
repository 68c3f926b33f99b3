#!/bin/bash
# how small can a graph be for the block path with write-back stores on XCD-exclusive rows?
mkdir -p gpurun_out
L=gpurun_out/r3_small_quality3.log; : > $L
for spec in "34 2 340 30" "128 2 1280 30" "512 2 5120 30" "1024 3 10240 30" "8192 5 81920 10"; do
  set -- $spec
  echo "== BA $1 x $2, $3 walks x $4 epochs" >> $L
  timeout 900 python scripts/quality_probe.py --nodes $1 --m $2 --walks $3 --epochs $4 --round-walks $3 \
     --modes atomic,write_back,blocks:1:8::st 2>&1 | grep -v amdgpu.ids >> $L
done
cat $L
