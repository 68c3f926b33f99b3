#!/bin/bash
# round 6, seventh GPU call: the evidence of the round -- rocprofv3 stats + calibrated PMC passes of
# the driver's command line (profiles/r06_*), the resident kernel's SQ counters, the TCC counters
# (atomic probe + resident kernel), the walk sampler's passes, a rank of 8 and BA 100 M
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_world.py -q > gpurun_out/r6/t7_world.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r6/t7_world.log
bash scripts/profile_bench.sh r06 > gpurun_out/r6/prof_r06.log 2>&1
bash scripts/resident_counters.sh > gpurun_out/r6/resident_counters.log 2>&1
bash scripts/atomic_counters.sh > gpurun_out/r6/atomic_counters.log 2>&1
bash scripts/profile_walks.sh r06_walks > gpurun_out/r6/prof_walks.log 2>&1
timeout 900 python bench.py --no-cpu-baseline --phantom-world 8 > gpurun_out/r6/bench7_phantom8.json 2> gpurun_out/r6/bench7_phantom8.err
timeout 1500 python bench.py --no-cpu-baseline --nodes 100000000 --steps 16 --warmup 8 > gpurun_out/r6/bench7_100m.json 2> gpurun_out/r6/bench7_100m.err
tail -5 gpurun_out/r6/t7_world.log
for f in gpurun_out/r6/bench7_phantom8.json gpurun_out/r6/bench7_100m.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"].get("frac"), d["roofline"].get("kernel_pairs_per_s"), d["config"]["parallelism"][-170:])
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
tail -3 gpurun_out/r6/prof_r06.log; tail -3 gpurun_out/r6/resident_counters.log | cut -c1-300
