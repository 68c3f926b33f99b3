#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_blocks.py -m gpu -q -s -k "shared_by_several" > gpurun_out/r3_shared_row.log 2>&1
grep "shared-row" gpurun_out/r3_shared_row.log
bash scripts/profile_bench.sh r03 > gpurun_out/r3_prof_r03.log 2>&1
tail -3 gpurun_out/r3_prof_r03.log
bash scripts/profile_bench.sh r03_100m --nodes 100000000 --steps 8 --warmup 8 > gpurun_out/r3_prof_r03_100m.log 2>&1
tail -3 gpurun_out/r3_prof_r03_100m.log
bash scripts/profile_bench.sh r03_cbow --model cbow > gpurun_out/r3_prof_r03_cbow.log 2>&1
tail -3 gpurun_out/r3_prof_r03_cbow.log
