#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_blocks.py -m gpu -q -k "2560 or small_graph" > gpurun_out/r3_gputests48.log 2>&1
tail -12 gpurun_out/r3_gputests48.log
