#!/bin/bash
# flakiness check: the GPU suite twice more on one box
mkdir -p gpurun_out
for i in 1 2; do
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r3_gputests71_$i.log 2>&1
  echo "run $i rc=$?"; tail -2 gpurun_out/r3_gputests71_$i.log
done
