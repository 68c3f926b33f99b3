#!/bin/bash
# traffic of one rank of eight (phantom world, no fabric): the figure the N = 8 bench line quotes
mkdir -p gpurun_out
bash scripts/profile_bench.sh r03_phantom8 --phantom-world 8 > gpurun_out/r3_prof_r03_phantom8.log 2>&1
tail -3 gpurun_out/r3_prof_r03_phantom8.log
grep -h '^{' gpurun_out/prof_r03_phantom8/stats.log | tail -1 | cut -c1-600
