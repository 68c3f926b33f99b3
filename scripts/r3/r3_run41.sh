#!/bin/bash
# CBOW on config 5's graph (BA 100 M / 1 B) on one GPU
mkdir -p gpurun_out
timeout 1500 python bench.py --model cbow --nodes 100000000 --steps 8 --warmup 2 --cpu-seconds 10 > gpurun_out/r3_bench41_cbow_100m.json 2> gpurun_out/r3_bench41_cbow_100m.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r3_bench41_cbow_100m.json") if l.startswith("{")][-1]); r=d["roofline"]
print("value %.4e %s ms/step %.1f frac %.3f kernel %s launch %.2f ms mem %s finite %s cpu %.3e"%(d["value"], d["unit"], d["ms_per_step"], r["frac"], r["kernel"], r["avg_launch_ms"], d["hbm_peak_gb"], d["finite"], d["cpu_baseline"]["value"]))
PY
tail -3 gpurun_out/r3_bench41_cbow_100m.err
