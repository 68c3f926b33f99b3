#!/bin/bash
mkdir -p gpurun_out
timeout 900 python scripts/soak.py 3000 > gpurun_out/r3_soak_3k.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r3_soak_3k.log | grep -v "^\[gn2v\]" | tail -8
timeout 900 python scripts/soak.py 30000 > gpurun_out/r3_soak_30k.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r3_soak_30k.log | grep -v "^\[gn2v\]" | tail -8
