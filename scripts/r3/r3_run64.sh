#!/bin/bash
# does the small-graph limit of the block path (2 560 nodes, measured at d = 128) hold for narrow rows?
mkdir -p gpurun_out
L=gpurun_out/r3_small_quality_d.log; : > $L
for spec in "2708 2 27080 30" "4096 4 40960 30" "8192 5 81920 10" "20000 5 200000 5"; do
  set -- $spec
  for d in 16 32 64; do
    echo "== BA $1 x $2, $3 walks x $4 epochs, d=$d" >> $L
    timeout 900 python scripts/quality_probe.py --nodes $1 --m $2 --walks $3 --epochs $4 --round-walks $3 --d $d \
       --modes atomic,blocks:1:8::st 2>&1 | grep -v amdgpu.ids | cut -c1-150 >> $L
  done
done
cat $L
