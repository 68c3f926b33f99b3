#!/bin/bash
mkdir -p gpurun_out
timeout 2000 python scripts/param_sweep.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_param_sweep.log
cat gpurun_out/r3_param_sweep.log
