#!/bin/bash
# same-box A/B: records of single-pair runs trained pair per group (ppg) vs the run-major loop only
mkdir -p gpurun_out
run() {
  tag=$1; shift
  GN2V_HIPCC_FLAGS="$*" python -c "from embiggen_amd import _lib; _lib.build(force=True)" || exit 1
  timeout 900 python bench.py --nodes 100000000 --steps 8 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench17_100m_$tag.json 2> gpurun_out/r3_bench17_100m_$tag.err
  timeout 600 python bench.py --steps 8 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench17_$tag.json 2> gpurun_out/r3_bench17_$tag.err
}
run ppg
run noppg -DGN2V_BLOCK_NO_PPG
run ppg_b
python -c "from embiggen_amd import _lib; _lib.build(force=True)"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_bench17*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); r=d["roofline"]
        print(f, "value %.3e"%d["value"], "frac %.3f"%r["frac"], "run %.2f"%r["mean_centre_run"], "launch %.2f ms"%r["avg_launch_ms"])
    except Exception as e: print(f, "failed", e)
PY
timeout 1200 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -m gpu -q > gpurun_out/r3_gputests17.log 2>&1
tail -4 gpurun_out/r3_gputests17.log
