#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_prediction_transformers.py -m gpu -q > gpurun_out/r3_gputests39.log 2>&1
tail -8 gpurun_out/r3_gputests39.log
