#!/bin/bash
# traffic of one rank of two and of four (phantom world): what the N = 2 / N = 4 bench lines quote
mkdir -p gpurun_out
for n in 2 4; do
  bash scripts/profile_bench.sh r03_phantom$n --phantom-world $n > gpurun_out/r3_prof_r03_phantom$n.log 2>&1
  tail -2 gpurun_out/r3_prof_r03_phantom$n.log
done
