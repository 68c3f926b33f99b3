#!/bin/bash
# where should a record switch to the pair-per-group path?  100 M nodes, one rank of 8 (runs of 1.4 pairs)
mkdir -p gpurun_out
for pct in 75 60 45; do
  GN2V_HIPCC_FLAGS="-DGN2V_PPG_MIN_PCT=$pct" python -c "from embiggen_amd import _lib; _lib.build(force=True)" || exit 1
  timeout 1200 python bench.py --nodes 100000000 --phantom-world 8 --steps 8 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench28_100m_phantom8_pct$pct.json 2> gpurun_out/r3_bench28_100m_phantom8_pct$pct.err
  timeout 900 python bench.py --steps 8 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench28_pct$pct.json 2> gpurun_out/r3_bench28_pct$pct.err
done
python -c "from embiggen_amd import _lib; _lib.build(force=True)"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_bench28*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); r=d["roofline"]
        print(f, "value %.3e"%d["value"], "frac %.3f"%r["frac"], "run %.2f"%r["mean_centre_run"], "launch %.2f ms"%r["avg_launch_ms"])
    except Exception as e: print(f, "failed", e)
PY
