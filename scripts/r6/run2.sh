#!/bin/bash
# round 6, second GPU call: the -m gpu suite after max_neighbours + the rounds-per-epoch rule +
# the regenerated block golden; walk-sampler rates by max_neighbours; the quality gates' numbers;
# one bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
timeout 1700 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_quality_gates.py > gpurun_out/r6/t2.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r6/t2.log
timeout 1500 python -m pytest tests/test_gpu_quality_gates.py -q -s > gpurun_out/r6/gates.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r6/gates.log
timeout 600 python scripts/typed_walk_probe.py > gpurun_out/r6/walk_rates.log 2>&1
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r6/bench2.json 2> gpurun_out/r6/bench2.err
timeout 600 python bench.py --no-cpu-baseline --max-neighbours 0 > gpurun_out/r6/bench2_exact.json 2> gpurun_out/r6/bench2_exact.err
tail -5 gpurun_out/r6/t2.log; grep -v "^\[\|amdgpu.ids" gpurun_out/r6/gates.log | tail -12; cat gpurun_out/r6/walk_rates.log | grep steps
for f in gpurun_out/r6/bench2.json gpurun_out/r6/bench2_exact.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], d["value"], d["ms_per_step"], d["walk_kernel_steps_per_s"], d["roofline"].get("frac"), d["config"]["parallelism"][-160:])
PY
done
