#!/bin/bash
# round 6, fourteenth GPU call: the committed evidence re-taken on the final tree
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
bash scripts/profile_bench.sh r06 > gpurun_out/r6/prof_r06_final.log 2>&1
bash scripts/resident_counters.sh > gpurun_out/r6/resident_counters_final.log 2>&1
bash scripts/atomic_counters.sh > gpurun_out/r6/atomic_counters_final.log 2>&1
timeout 600 python bench.py > gpurun_out/r6/bench14.json 2> gpurun_out/r6/bench14.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6/bench14.json"))
print(d["value"], d["ms_per_step"], d["walk_kernel_steps_per_s"], d["roofline"].get("frac"), d["roofline"].get("frac_hbm"), d.get("first_fit_s"), d["cpu_baseline"]["value"])
PY
