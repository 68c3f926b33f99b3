#!/bin/bash
# CBOW kernel A/B on one box: output-row loads issued before / after the window mean, VGPR cap
mkdir -p gpurun_out
run() {
  tag=$1; shift
  GN2V_HIPCC_FLAGS="$*" python -c "from embiggen_amd import _lib; _lib.build(force=True)" || exit 1
  for i in 1 2; do
    timeout 600 python bench.py --model cbow --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3_cbow_ab_$tag.$i.json 2> gpurun_out/r3_cbow_ab_$tag.$i.err
  done
}
run late127 -DGN2V_CBOW_EARLY_ISSUE=0
run early153
run early128 -DGN2V_CBOW_MIN_BLOCKS=4
python -c "from embiggen_amd import _lib; _lib.build(force=True)"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_cbow_ab_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); r=d["roofline"]
        print(f, "value %.3e"%d["value"], "frac %.3f"%r["frac"], "launch %.2f ms"%r["avg_launch_ms"])
    except Exception as e: print(f, "failed", e)
PY
