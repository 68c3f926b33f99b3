#!/bin/bash
# round 6, call 37: gn2v_walks_strided -- a Node2VecSequence batch in one launch
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_gpu_walks.py tests/test_gpu_typed_walks.py tests/test_gpu_twins.py tests/test_gpu_api.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python scripts/sequence_probe.py 2>&1 | tail -6
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['frac'], d['walk_kernel_steps_per_s'])"
