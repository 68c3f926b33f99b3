#!/bin/bash
# round 6, call 36: eight processes over gloo sharing the box's one GPU through bench.py --gpus 8 (the driver's
# launch line with --backend gloo --share-device): the multi-process path end to end on the final tree
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
timeout 1200 python bench.py --gpus 8 --backend gloo --share-device --nodes 2000000 --walks 32768 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r6/gloo8.json 2> gpurun_out/r6/gloo8.err
echo "rc=$?"; tail -c 1500 gpurun_out/r6/gloo8.json; tail -3 gpurun_out/r6/gloo8.err
