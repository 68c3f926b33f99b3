#!/bin/bash
# BASELINE config 2's shape (ogbn-arxiv: 169 343 nodes / 1.17 M edges, p = 0.5, q = 2) through bench.py
mkdir -p gpurun_out
bash scripts/profile_bench.sh r03_arxiv --nodes 169343 --m 7 --return-weight 2.0 --explore-weight 0.5 --walks 169343 > gpurun_out/r3_prof_r03_arxiv.log 2>&1
tail -3 gpurun_out/r3_prof_r03_arxiv.log
grep -h '^{' gpurun_out/prof_r03_arxiv/stats.log | tail -1 | cut -c1-400
