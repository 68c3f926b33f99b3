#!/bin/bash
mkdir -p gpurun_out
run() {
  tag=$1; shift
  GN2V_HIPCC_FLAGS="$*" python -c "from embiggen_amd import _lib; _lib.build(force=True)" || exit 1
  for i in 1 2; do
    timeout 600 python bench.py --model cbow --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3_bench13_cbow_$tag.$i.json 2> gpurun_out/r3_bench13_cbow_$tag.$i.err
  done
}
run waves3
run waves4 -DGN2V_CBOW_LAZY_MIN_BLOCKS=4
python -c "from embiggen_amd import _lib; _lib.build(force=True)"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_bench13_*.json")):
    d=json.loads([l for l in open(f) if l.startswith("{")][-1]); r=d["roofline"]
    print(f, "value %.3e frac %.3f launch %.2f ms finite %s"%(d["value"], r["frac"], r["avg_launch_ms"], d["finite"]))
PY
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r3_gputests13.log 2>&1
tail -6 gpurun_out/r3_gputests13.log
