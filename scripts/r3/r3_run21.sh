#!/bin/bash
mkdir -p gpurun_out
for tag in full nofull full_b nofull_b; do
  unset GN2V_BLOCK_NO_FULL; case $tag in nofull*) export GN2V_BLOCK_NO_FULL=1;; esac
  timeout 600 python bench.py --model cbow --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3_bench21_cbow_$tag.json 2> gpurun_out/r3_bench21_cbow_$tag.err
done
unset GN2V_BLOCK_NO_FULL
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_bench21*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); r=d["roofline"]
        print(f, "value %.3e"%d["value"], "frac %.3f"%r["frac"], "launch %.2f ms"%r["avg_launch_ms"])
    except Exception as e: print(f, "failed", e)
PY
