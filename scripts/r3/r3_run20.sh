#!/bin/bash
# same-box A/B: block kernel with the row stride as a compile-time constant (FULL, ships) vs the general one
mkdir -p gpurun_out
for tag in full nofull full_b nofull_b; do
  unset GN2V_BLOCK_NO_FULL; case $tag in nofull*) export GN2V_BLOCK_NO_FULL=1;; esac
  timeout 600 python bench.py --steps 8 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench20_$tag.json 2> gpurun_out/r3_bench20_$tag.err
  timeout 900 python bench.py --nodes 100000000 --steps 8 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench20_100m_$tag.json 2> gpurun_out/r3_bench20_100m_$tag.err
done
unset GN2V_BLOCK_NO_FULL
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_bench20*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); r=d["roofline"]
        print(f, "value %.3e"%d["value"], "frac %.3f"%r["frac"], "launch %.2f ms"%r["avg_launch_ms"])
    except Exception as e: print(f, "failed", e)
PY
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r3_gputests20.log 2>&1
tail -4 gpurun_out/r3_gputests20.log
