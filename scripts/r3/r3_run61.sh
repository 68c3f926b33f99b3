#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_api.py tests/test_gpu_bench_contract.py -m gpu -q > gpurun_out/r3_gputests61.log 2>&1
tail -12 gpurun_out/r3_gputests61.log
timeout 600 python scripts/soak.py 30000 Node2VecCBOWEnsmallen 2>&1 | grep -v amdgpu.ids | grep -v "^\[gn2v\]" | tail -2
