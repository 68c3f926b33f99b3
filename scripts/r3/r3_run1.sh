#!/bin/bash
# round 3, GPU run 1: the whole GPU suite, then the default bench and the 100 M-node graph
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputests1.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_gputests1.log
tail -5 gpurun_out/r3_gputests1.log
timeout 600 python bench.py --steps 16 --warmup 4 > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err
tail -c 600 gpurun_out/r3_bench1.json
timeout 900 python bench.py --nodes 100000000 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r3_bench1_100m.json 2> gpurun_out/r3_bench1_100m.err
tail -c 600 gpurun_out/r3_bench1_100m.json
