#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_blocks.py -m gpu -q -k "pair_per_group or collision_free or long_stretches" > gpurun_out/r3_gputests33.log 2>&1
tail -12 gpurun_out/r3_gputests33.log
