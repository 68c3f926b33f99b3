#!/bin/bash
# round 6, eighth GPU call: the walk sampler after RowView went by value (tests, rocprofv3 passes),
# a rank of 8 with rounds of at least 2^19 walks, the driver's line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_walks.py tests/test_gpu_typed_walks.py tests/test_gpu_world.py -q > gpurun_out/r6/t8.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r6/t8.log
bash scripts/profile_walks.sh r06_walks > gpurun_out/r6/prof_walks8.log 2>&1
timeout 900 python bench.py --no-cpu-baseline --phantom-world 8 > gpurun_out/r6/bench8_phantom8.json 2> gpurun_out/r6/bench8_phantom8.err
timeout 600 python bench.py > gpurun_out/r6/bench8.json 2> gpurun_out/r6/bench8.err
tail -3 gpurun_out/r6/t8.log
for f in gpurun_out/r6/bench8_phantom8.json gpurun_out/r6/bench8.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1], d["value"], d["ms_per_step"], d["walk_kernel_steps_per_s"], d["roofline"].get("frac"), d["roofline"].get("kernel_pairs_per_s"), d["config"]["parallelism"][-170:])
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
