#!/bin/bash
# small graphs through the block path with explicit store flavours on the XCD-exclusive rows
mkdir -p gpurun_out
L=gpurun_out/r3_small_quality2.log; : > $L
for spec in "2708 2 27080 10" "20000 5 200000 5" "50000 7 500000 3"; do
  set -- $spec
  echo "== BA $1 x $2, $3 walks x $4 epochs" >> $L
  timeout 900 python scripts/quality_probe.py --nodes $1 --m $2 --walks $3 --epochs $4 --round-walks $3 \
     --modes write_back,blocks:1:8::la,blocks:1:8::st,blocks:1:1::st 2>&1 | grep -v amdgpu.ids >> $L
done
cat $L
