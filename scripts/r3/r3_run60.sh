#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python scripts/cbow_store_graphs.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_cbow_store_graphs.log
cat gpurun_out/r3_cbow_store_graphs.log
