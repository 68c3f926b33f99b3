#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python scripts/cbow_batch_quality.py 200000 2097152 > gpurun_out/r3_cbow_batch_quality.log 2>&1
tail -15 gpurun_out/r3_cbow_batch_quality.log
for rec in 24 32; do
  timeout 900 python bench.py --nodes 100000000 --steps 8 --warmup 8 --no-cpu-baseline --record $rec > gpurun_out/r3_bench4_100m_rec$rec.json 2> gpurun_out/r3_bench4_100m_rec$rec.err
done
timeout 600 python bench.py --steps 8 --warmup 8 --no-cpu-baseline --record 32 > gpurun_out/r3_bench4_rec32.json 2> gpurun_out/r3_bench4_rec32.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_bench4*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); r=d["roofline"]
        print(f, "value %.3e"%d["value"], "frac %.3f"%r["frac"], "sched %.3f"%r["frac_scheduled"], "run %.2f"%r["mean_centre_run"], "launch %.1f ms"%r["avg_launch_ms"], "mem %.0f GB"%d["hbm_peak_gb"]["torch_allocated"])
    except Exception as e: print(f, "failed", e)
PY
