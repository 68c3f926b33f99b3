#!/bin/bash
# whole default-parameter fits through the public classes, BA 1 M nodes (30 epochs x 10 iterations for
# SkipGram / CBOW: 3.75e11 pairs / 3.84e10 centres)
mkdir -p gpurun_out
timeout 2400 python scripts/soak.py 1000000 > gpurun_out/r3_soak_1m.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r3_soak_1m.log | tail -8
