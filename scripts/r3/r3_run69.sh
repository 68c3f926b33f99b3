#!/bin/bash
# BASELINE config 3's shape (ogbn-products: 2 449 029 nodes / 61.9 M edges) on one GPU through bench.py
mkdir -p gpurun_out
bash scripts/profile_bench.sh r03_products --nodes 2449029 --m 25 > gpurun_out/r3_prof_r03_products.log 2>&1
tail -3 gpurun_out/r3_prof_r03_products.log
grep -h '^{' gpurun_out/prof_r03_products/stats.log | tail -1 | cut -c1-400
