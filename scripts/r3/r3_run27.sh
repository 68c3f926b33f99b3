#!/bin/bash
# the committed profiles, re-taken with the final build
mkdir -p gpurun_out
bash scripts/profile_bench.sh r03 > gpurun_out/r3_prof_r03.log 2>&1
tail -2 gpurun_out/r3_prof_r03.log
bash scripts/profile_bench.sh r03_100m --nodes 100000000 --steps 8 --warmup 8 > gpurun_out/r3_prof_r03_100m.log 2>&1
tail -2 gpurun_out/r3_prof_r03_100m.log
bash scripts/profile_bench.sh r03_cbow --model cbow > gpurun_out/r3_prof_r03_cbow.log 2>&1
tail -2 gpurun_out/r3_prof_r03_cbow.log
