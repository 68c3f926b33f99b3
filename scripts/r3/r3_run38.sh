#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_prediction_transformers.py tests/test_transformers.py tests/test_gpu_glove.py -m gpu -q > gpurun_out/r3_gputests38.log 2>&1
tail -15 gpurun_out/r3_gputests38.log
timeout 600 python tests/gpu_check.py 2>&1 | grep -v amdgpu.ids | tail -12
