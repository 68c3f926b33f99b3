#!/bin/bash
# block path from 2 560 nodes up: GPU suite, smoke, small default fits
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r3_gputests47.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_gputests47.log
tail -15 gpurun_out/r3_gputests47.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1200 python scripts/small_fits.py > gpurun_out/r3_small_fits2.log 2>&1
grep -v amdgpu.ids gpurun_out/r3_small_fits2.log | grep -v "^\[gn2v\]" | tail -12
