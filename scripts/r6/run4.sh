#!/bin/bash
# round 6, fourth GPU call: the whole -m gpu suite (header split, CBOW full-size tests, walk
# kernels with the SUB instantiation), rounds-per-epoch A/B at config 3, walk rate by
# max_neighbours through the bench, CBOW bench line + rocprofv3 passes
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
timeout 1700 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_quality_gates.py > gpurun_out/r6/t4.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r6/t4.log
for cfg in "64 16384" "128 8192"; do
  set -- $cfg
  GN2V_ROUNDS_PER_EPOCH=$1 GN2V_ROUND_MIN_WALKS=$2 timeout 900 python -m pytest tests/test_gpu_quality_gates.py -q -s -k config3 > gpurun_out/r6/gates4_r$1.log 2>&1
done
GN2V_RESIDENT_MIN_NODES=1000000000 timeout 900 python -m pytest tests/test_gpu_quality_gates.py -q -s -k config3 > gpurun_out/r6/gates4_xcd.log 2>&1
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r6/bench4.json 2> gpurun_out/r6/bench4.err
timeout 600 python bench.py --no-cpu-baseline --max-neighbours 0 > gpurun_out/r6/bench4_exact.json 2> gpurun_out/r6/bench4_exact.err
timeout 900 python bench.py --model cbow > gpurun_out/r6/bench4_cbow.json 2> gpurun_out/r6/bench4_cbow.err
bash scripts/profile_bench.sh r06_cbow --model cbow > gpurun_out/r6/prof_cbow.log 2>&1
tail -4 gpurun_out/r6/t4.log
grep -h "default (resident\|passed\|failed" gpurun_out/r6/gates4_*.log
for f in gpurun_out/r6/bench4.json gpurun_out/r6/bench4_exact.json gpurun_out/r6/bench4_cbow.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], d["value"], d["unit"], d["ms_per_step"], d["walk_kernel_steps_per_s"], d["roofline"].get("frac"), d.get("first_fit_s"), d.get("cpu_baseline",{}).get("value"))
PY
done
