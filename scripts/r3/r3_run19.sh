#!/bin/bash
# one rank of N without the fabric (bench.py --phantom-world N), final kernel
mkdir -p gpurun_out
for w in 2 4 8; do
  timeout 900 python bench.py --phantom-world $w --steps 8 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench19_phantom$w.json 2> gpurun_out/r3_bench19_phantom$w.err
done
timeout 1200 python bench.py --nodes 100000000 --phantom-world 8 --steps 8 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench19_100m_phantom8.json 2> gpurun_out/r3_bench19_100m_phantom8.err
timeout 600 python bench.py --steps 8 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench19_one.json 2> gpurun_out/r3_bench19_one.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_bench19*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); r=d["roofline"]
        print(f, "value %.3e"%d["value"], "frac %.3f"%r["frac"], "sched %.3f"%r["frac_scheduled"], "run %.2f"%r["mean_centre_run"], "launch %.1f ms"%r["avg_launch_ms"], "mem %.0f/%.0f"%(d["hbm_peak_gb"]["torch_allocated"], d["hbm_peak_gb"]["torch_reserved"]), d.get("phantom",{}).get("hop_copies",{}).get("mean_ms"))
    except Exception as e: print(f, "failed", e); print(open(f.replace(".json",".err")).read()[-800:])
PY
