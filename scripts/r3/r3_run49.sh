#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r3_cbow_small.log; : > $L
for spec in "2708 2 30" "8192 5 10" "20000 5 5" "50000 7 3"; do
  timeout 600 python scripts/cbow_small_probe.py $spec 2>&1 | grep -v amdgpu.ids >> $L
done
cat $L
