#!/bin/bash
# round 6, ninth GPU call: the counted scatter (no radix sort for resident plans): the whole suite,
# the bench line, a rank of 8, BA 100 M, the rocprofv3 stats of the driver's command line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
timeout 1700 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_quality_gates.py > gpurun_out/r6/t9.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r6/t9.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r6/bench9.json 2> gpurun_out/r6/bench9.err
timeout 900 python bench.py --no-cpu-baseline --phantom-world 8 > gpurun_out/r6/bench9_phantom8.json 2> gpurun_out/r6/bench9_phantom8.err
timeout 1500 python bench.py --no-cpu-baseline --nodes 100000000 --steps 16 --warmup 8 > gpurun_out/r6/bench9_100m.json 2> gpurun_out/r6/bench9_100m.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r06s/stats" -o stats -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/r6/prof9_stats.log" 2>&1
cd "$GRAFT_REPO_ROOT"
tail -6 gpurun_out/r6/t9.log
for f in gpurun_out/r6/bench9.json gpurun_out/r6/bench9_phantom8.json gpurun_out/r6/bench9_100m.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"].get("frac"), d["roofline"].get("kernel_pairs_per_s"), d["hbm_peak_gb"], d["config"]["parallelism"][-120:])
except Exception as e: print(sys.argv[1], "FAILED", e, open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
done
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_r06s/stats/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]: print(r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e6, r["Percentage"])
PY
