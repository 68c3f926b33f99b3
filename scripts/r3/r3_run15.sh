#!/bin/bash
# same-box A/B of the block kernel: lean scoring (ships) vs per-round duplicate analysis; 5 blocks per CU
mkdir -p gpurun_out
run() {
  tag=$1; shift
  GN2V_HIPCC_FLAGS="$*" python -c "from embiggen_amd import _lib; _lib.build(force=True)" || exit 1
  timeout 600 python bench.py --steps 8 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench15_$tag.json 2> gpurun_out/r3_bench15_$tag.err
  timeout 900 python bench.py --nodes 100000000 --steps 8 --warmup 8 --no-cpu-baseline > gpurun_out/r3_bench15_100m_$tag.json 2> gpurun_out/r3_bench15_100m_$tag.err
}
run lean
run checked -DGN2V_BLOCK_SERIALISE
run lean5 -DGN2V_BLOCK_MIN_BLOCKS=5
run lean_b
python -c "from embiggen_amd import _lib; _lib.build(force=True)"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_bench15*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); r=d["roofline"]
        print(f, "value %.3e"%d["value"], "frac %.3f"%r["frac"], "sched %.3f"%r["frac_scheduled"], "launch %.1f ms"%r["avg_launch_ms"])
    except Exception as e: print(f, "failed", e)
PY
timeout 900 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_configs.py -m gpu -q > gpurun_out/r3_gputests15.log 2>&1
tail -4 gpurun_out/r3_gputests15.log
