#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_configs.py -m gpu -q -k "stripes or quality or simulated or sliced or config4_block" > gpurun_out/r3_gputests7.log 2>&1
tail -5 gpurun_out/r3_gputests7.log
# link quality, BA 1 M, 3 epochs: default (stores for single-run centres) vs atomics for every run
for extra in "" "--central-atomic"; do
  timeout 900 python scripts/quality_probe.py --nodes 1000000 --epochs 3 --modes write_through,blocks:3:8,blocks:1:8 $extra > gpurun_out/r3_quality7$extra.log 2>&1
  tail -4 gpurun_out/r3_quality7$extra.log
done
bash scripts/cbow_ab.sh 2>&1 | tail -8
