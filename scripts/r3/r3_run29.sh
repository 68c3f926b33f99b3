#!/bin/bash
# experiment: flagged (high-degree) rows stay in L2, all other sample rows are streamed (nt loads / stores)
mkdir -p gpurun_out
timeout 600 python bench.py --steps 8 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench29_base.json 2> gpurun_out/r3_bench29_base.err
for band in 16:0 15:0 14:0 13:0; do
  tag=$(echo $band | tr ':' '_')
  GN2V_BLOCK_RETAIN=1 timeout 600 python bench.py --steps 8 --warmup 8 --no-cpu-baseline --hot-band $band > gpurun_out/r3_bench29_retain_$tag.json 2> gpurun_out/r3_bench29_retain_$tag.err
done
timeout 600 python bench.py --steps 8 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench29_base_b.json 2> gpurun_out/r3_bench29_base_b.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_bench29*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); r=d["roofline"]
        print(f, "value %.3e"%d["value"], "frac %.3f"%r["frac"], "launch %.2f ms"%r["avg_launch_ms"], d["finite"])
    except Exception as e: print(f, "failed", e); print(open(f.replace(".json",".err")).read()[-500:])
PY
