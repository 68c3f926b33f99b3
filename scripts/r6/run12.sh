#!/bin/bash
# round 6, twelfth GPU call: after pruning eight environment switches -- the whole suite with the
# gates (displacement ratios added), smoke(), the driver's bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r6/t12.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r6/t12.log
timeout 1500 python -m pytest tests/test_gpu_quality_gates.py -q -s > gpurun_out/r6/gates12.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6/smoke12.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r6/smoke12.log
timeout 600 python bench.py > gpurun_out/r6/bench12.json 2> gpurun_out/r6/bench12.err
tail -4 gpurun_out/r6/t12.log; grep -h "default (resident\|bench graph:\|passed\|failed" gpurun_out/r6/gates12.log | cut -c1-700; tail -2 gpurun_out/r6/smoke12.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6/bench12.json"))
print(d["value"], d["ms_per_step"], d["walk_kernel_steps_per_s"], d["roofline"].get("frac"), d["roofline"].get("frac_hbm"), d.get("first_fit_s"), d["cpu_baseline"]["value"])
PY
