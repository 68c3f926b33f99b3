#!/bin/bash
# round 6, sixth GPU call: the C world loop against the Python trainer, the quality gates under
# the final rounds rule, the whole suite, one bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_world.py -q -x > gpurun_out/r6/t6_world.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r6/t6_world.log
timeout 1500 python -m pytest tests/test_gpu_quality_gates.py -q -s > gpurun_out/r6/gates6.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r6/gates6.log
timeout 1700 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_quality_gates.py --deselect tests/test_gpu_world.py > gpurun_out/r6/t6.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r6/t6.log
timeout 600 python bench.py > gpurun_out/r6/bench6.json 2> gpurun_out/r6/bench6.err
tail -25 gpurun_out/r6/t6_world.log; grep -h "default (resident\|bench graph:\|passed\|failed" gpurun_out/r6/gates6.log; tail -4 gpurun_out/r6/t6.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6/bench6.json"))
print(d["value"], d["ms_per_step"], d["walk_kernel_steps_per_s"], d["roofline"].get("frac"), d["roofline"].get("frac_hbm"), d.get("first_fit_s"), d["cpu_baseline"]["value"], d["config"]["parallelism"][-150:])
PY
