#!/bin/bash
# round 6, third GPU call: walk sampler after the cheaper sub-sample hash (tests + rates), the
# quality gates with their noise floor (and with 32 rounds an epoch), the L2 atomic counters
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_walks.py tests/test_gpu_typed_walks.py tests/test_gpu_resident.py tests/test_gpu_api.py -q -x > gpurun_out/r6/t3.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r6/t3.log
timeout 600 python scripts/typed_walk_probe.py > gpurun_out/r6/walk_rates3.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_quality_gates.py -q -s > gpurun_out/r6/gates3.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r6/gates3.log
GN2V_ROUNDS_PER_EPOCH=32 timeout 900 python -m pytest tests/test_gpu_quality_gates.py -q -s -k config3 > gpurun_out/r6/gates3_r32.log 2>&1
GN2V_ROUNDS_PER_EPOCH=64 timeout 900 python -m pytest tests/test_gpu_quality_gates.py -q -s -k config3 > gpurun_out/r6/gates3_r64.log 2>&1
bash scripts/atomic_counters.sh > gpurun_out/r6/atomic_counters.log 2>&1
tail -3 gpurun_out/r6/t3.log; grep steps gpurun_out/r6/walk_rates3.log
grep -h "default (resident\|bench graph:\|passed\|failed" gpurun_out/r6/gates3.log gpurun_out/r6/gates3_r32.log gpurun_out/r6/gates3_r64.log
tail -c 3000 gpurun_out/atomic_counters.json
