#!/bin/bash
mkdir -p gpurun_out
GN2V_HIPCC_FLAGS="-DGN2V_CBOW_LAZY_MIN_BLOCKS=5" python -c "from embiggen_amd import _lib; _lib.build(force=True)"
timeout 600 python bench.py --model cbow --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3_bench14_cbow_waves5.json 2> gpurun_out/r3_bench14_cbow_waves5.err
python -c "from embiggen_amd import _lib; _lib.build(force=True)"
timeout 600 python bench.py --model cbow --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3_bench14_cbow_waves4.json 2> gpurun_out/r3_bench14_cbow_waves4.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_bench14_*.json")):
    d=json.loads([l for l in open(f) if l.startswith("{")][-1]); r=d["roofline"]
    print(f, "value %.3e frac %.3f launch %.2f ms finite %s"%(d["value"], r["frac"], r["avg_launch_ms"], d["finite"]))
PY
# link quality of the lazy kernel vs the cached one (BA 200 k, the probe's one-trainer line) and the whole fit
GN2V_CBOW_LAZY=1 timeout 600 python scripts/cbow_batch_quality.py 200000 2097152 2>&1 | head -2 | tail -1
GN2V_CBOW_LAZY=0 timeout 600 python scripts/cbow_batch_quality.py 200000 2097152 2>&1 | head -2 | tail -1
bash scripts/profile_bench.sh r03_cbow --model cbow > gpurun_out/r3_prof_r03_cbow.log 2>&1
tail -2 gpurun_out/r3_prof_r03_cbow.log
