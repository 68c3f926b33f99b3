#!/bin/bash
# BASELINE config 1's shape (Cora: 2 708 nodes) through bench.py, with the rocprofv3 passes
mkdir -p gpurun_out
bash scripts/profile_bench.sh r03_cora --nodes 2708 --m 2 --walks 27080 > gpurun_out/r3_prof_r03_cora.log 2>&1
tail -3 gpurun_out/r3_prof_r03_cora.log
grep -h '^{' gpurun_out/prof_r03_cora/stats.log | tail -1 | cut -c1-900
