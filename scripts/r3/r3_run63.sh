#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -k "cbow" > gpurun_out/r3_gputests63.log 2>&1
tail -12 gpurun_out/r3_gputests63.log
