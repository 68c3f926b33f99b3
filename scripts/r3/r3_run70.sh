#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python scripts/soak.py 100000 > gpurun_out/r3_soak_100k.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r3_soak_100k.log | grep -v "^\[gn2v\]" | tail -8
