#!/bin/bash
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r3_gputests6.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_gputests6.log
tail -6 gpurun_out/r3_gputests6.log
for extra in "" "--reserve-cus 1"; do
  tag=$(echo $extra | tr -d ' -')
  timeout 900 python bench.py --phantom-world 8 --steps 8 --warmup 8 --no-cpu-baseline $extra > gpurun_out/r3_bench6_phantom8$tag.json 2> gpurun_out/r3_bench6_phantom8$tag.err
  timeout 900 python bench.py --steps 8 --warmup 8 --no-cpu-baseline $extra > gpurun_out/r3_bench6_one$tag.json 2> gpurun_out/r3_bench6_one$tag.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_bench6*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); r=d["roofline"]
        print(f, "value %.3e"%d["value"], "frac %.3f"%r["frac"], "sched %.3f"%r["frac_scheduled"], "run %.2f"%r["mean_centre_run"], "launch %.1f ms"%r["avg_launch_ms"], "mem %.0f"%d["hbm_peak_gb"]["torch_allocated"], d.get("phantom",{}).get("hop_copies"), d["config"].get("active_cus_per_xcd"))
    except Exception as e: print(f, "failed", e); print(open(f.replace(".json",".err")).read()[-1500:])
PY
