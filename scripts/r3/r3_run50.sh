#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "block_path" > gpurun_out/r3_gputests50.log 2>&1
tail -25 gpurun_out/r3_gputests50.log
