#!/bin/bash
# one GPU: next group prepared on the side stream while one trains (--overlap on) vs in line
mkdir -p gpurun_out
for tag in inline overlap inline_b overlap_b; do
  ov=off; case $tag in overlap*) ov=on;; esac
  timeout 900 python bench.py --overlap $ov --steps 16 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench26_$tag.json 2> gpurun_out/r3_bench26_$tag.err
done
timeout 900 python bench.py --nodes 100000000 --overlap on --steps 8 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench26_100m_overlap.json 2> gpurun_out/r3_bench26_100m_overlap.err
timeout 900 python bench.py --nodes 100000000 --overlap off --steps 8 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench26_100m_inline.json 2> gpurun_out/r3_bench26_100m_inline.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_bench26*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); r=d["roofline"]
        print(f, "value %.3e"%d["value"], "ms/step %.1f"%d["ms_per_step"], "frac %.3f"%r["frac"], "launch %.2f ms"%r["avg_launch_ms"], "mem %.0f"%d["hbm_peak_gb"]["torch_allocated"])
    except Exception as e: print(f, "failed", e)
PY
