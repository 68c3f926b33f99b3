#!/bin/bash
# same-box A/B: group reduction by permlane swaps (ships) vs ds_bpermute; 5 blocks per CU; then the whole GPU suite
mkdir -p gpurun_out
run() {
  tag=$1; shift
  GN2V_HIPCC_FLAGS="$*" python -c "from embiggen_amd import _lib; _lib.build(force=True)" || exit 1
  timeout 600 python bench.py --steps 8 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench16_$tag.json 2> gpurun_out/r3_bench16_$tag.err
  timeout 900 python bench.py --nodes 100000000 --steps 8 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench16_100m_$tag.json 2> gpurun_out/r3_bench16_100m_$tag.err
  timeout 600 python bench.py --model cbow --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3_bench16_cbow_$tag.json 2> gpurun_out/r3_bench16_cbow_$tag.err
}
run swap
run bperm -DGN2V_REDUCE_BPERMUTE
run swap5 -DGN2V_BLOCK_MIN_BLOCKS=5
python -c "from embiggen_amd import _lib; _lib.build(force=True)"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_bench16*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); r=d["roofline"]
        print(f, "value %.3e"%d["value"], "frac %.3f"%r["frac"], "launch %.2f ms"%r["avg_launch_ms"])
    except Exception as e: print(f, "failed", e)
PY
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r3_gputests16.log 2>&1
tail -4 gpurun_out/r3_gputests16.log
