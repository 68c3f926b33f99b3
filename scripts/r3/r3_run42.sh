#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python scripts/small_fits.py > gpurun_out/r3_small_fits.log 2>&1
grep -v amdgpu.ids gpurun_out/r3_small_fits.log | tail -12
