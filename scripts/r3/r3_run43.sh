#!/bin/bash
# small graphs (below 2^16 nodes: atomic walk-ordered kernel today) through the block path with
# L2-local atomics on XCD-exclusive contextual rows: throughput and link AUROC on identical walks
mkdir -p gpurun_out
L=gpurun_out/r3_small_quality.log; : > $L
for spec in "2708 2 27080 10" "20000 5 200000 5" "50000 7 500000 3"; do
  set -- $spec
  echo "== BA $1 x $2, $3 walks x $4 epochs" >> $L
  timeout 900 python scripts/quality_probe.py --nodes $1 --m $2 --walks $3 --epochs $4 --round-walks $3 \
     --modes atomic,write_through,blocks:1:8::la,blocks:1:8,blocks:1:1 2>&1 | grep -v amdgpu.ids >> $L
  timeout 900 python scripts/quality_probe.py --nodes $1 --m $2 --walks $3 --epochs $4 --round-walks $3 \
     --central-atomic --modes blocks:1:8::la 2>&1 | grep -v amdgpu.ids | sed 's/^/central-atomic /' >> $L
done
cat $L
