#!/bin/bash
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r3_gputests8.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_gputests8.log
tail -6 gpurun_out/r3_gputests8.log
timeout 600 python bench.py --model cbow --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3_bench8_cbow.json 2> gpurun_out/r3_bench8_cbow.err
timeout 600 python bench.py --steps 8 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench8.json 2> gpurun_out/r3_bench8.err
timeout 900 python bench.py --nodes 100000000 --steps 8 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench8_100m.json 2> gpurun_out/r3_bench8_100m.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_bench8*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); r=d["roofline"]
        print(f, "value %.3e"%d["value"], "frac %.3f"%r["frac"], "sched %.3f"%r["frac_scheduled"], "run %.2f"%r["mean_centre_run"], "launch %.1f ms"%r["avg_launch_ms"])
    except Exception as e: print(f, "failed", e)
PY
