#!/bin/bash
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r3_gputests2.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_gputests2.log
tail -8 gpurun_out/r3_gputests2.log
for extra in "" "--central-atomic"; do
  timeout 600 python bench.py --steps 8 --warmup 8 --no-cpu-baseline $extra > gpurun_out/r3_bench2$extra.json 2> gpurun_out/r3_bench2$extra.err
  tail -c 300 gpurun_out/r3_bench2$extra.json | head -c 10; echo
done
for extra in "" "--central-atomic"; do
  timeout 900 python bench.py --nodes 100000000 --steps 8 --warmup 8 --no-cpu-baseline $extra > gpurun_out/r3_bench2_100m$extra.json 2> gpurun_out/r3_bench2_100m$extra.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_bench2*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); r=d["roofline"]
        print(f, "value %.3e"%d["value"], "frac %.3f"%r["frac"], "sched %.3f"%r["frac_scheduled"], "run %.2f"%r["mean_centre_run"], "launch %.1f ms"%r["avg_launch_ms"], "mem %.0f GB"%d["hbm_peak_gb"]["torch_allocated"])
    except Exception as e: print(f, "failed", e)
PY
