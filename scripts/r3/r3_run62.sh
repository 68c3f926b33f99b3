#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python scripts/cbow_threshold_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_cbow_threshold.log
cat gpurun_out/r3_cbow_threshold.log
