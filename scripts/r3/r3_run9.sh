#!/bin/bash
# same-box A/B: run cap 16 (ships) vs 32 (= none at records of 32)
mkdir -p gpurun_out
run() {
  tag=$1; shift
  GN2V_HIPCC_FLAGS="$*" python -c "from embiggen_amd import _lib; _lib.build(force=True)" || exit 1
  for i in 1 2; do
    timeout 600 python bench.py --steps 8 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench9_$tag.$i.json 2> gpurun_out/r3_bench9_$tag.$i.err
  done
  timeout 900 python bench.py --nodes 100000000 --steps 8 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench9_100m_$tag.json 2> gpurun_out/r3_bench9_100m_$tag.err
}
run cap16
run cap32 -DGN2V_MAX_RUN=32
run cap16b
python -c "from embiggen_amd import _lib; _lib.build(force=True)"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_bench9*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); r=d["roofline"]
        print(f, "value %.3e"%d["value"], "frac %.3f"%r["frac"], "sched %.3f"%r["frac_scheduled"], "run %.2f"%r["mean_centre_run"], "launch %.1f ms"%r["avg_launch_ms"])
    except Exception as e: print(f, "failed", e)
PY
