#!/bin/bash
# CBOW lazy-window kernel at d = 256 / 512 / 1024 (CH = 4 / 8 / 16): register caps, same box
mkdir -p gpurun_out
L=gpurun_out/r3_cbow_wide_ab.log; : > $L
run() {
  tag=$1; shift
  GN2V_HIPCC_FLAGS="$*" python -c "from embiggen_amd import _lib; _lib.build(force=True)" || exit 1
  for d in 200 256 512 1024; do
    timeout 600 python bench.py --model cbow --d $d --nodes 1000000 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r3_cbow_wide_$tag.$d.json 2>/dev/null
    python - "$tag" "$d" >> $L <<'PY'
import json, sys
tag, d = sys.argv[1], sys.argv[2]
try:
    l = json.loads([x for x in open(f"gpurun_out/r3_cbow_wide_{tag}.{d}.json") if x.startswith("{")][-1]); r = l["roofline"]
    print(f"{tag:22s} d={d:5s} {l['value']:.3e} centres/s frac {r['frac']:.3f} launch {r['avg_launch_ms']:.2f} ms finite {l['finite']}")
except Exception as e:
    print(tag, d, "FAILED", e)
PY
  done
}
run cap4_all -DGN2V_CBOW_LAZY_MIN_BLOCKS_CH4=4 -DGN2V_CBOW_LAZY_MIN_BLOCKS_CH8=4 -DGN2V_CBOW_LAZY_MIN_BLOCKS_CH16=4
run cap3_2_1
run cap2_2_2 -DGN2V_CBOW_LAZY_MIN_BLOCKS_CH4=2 -DGN2V_CBOW_LAZY_MIN_BLOCKS_CH8=2 -DGN2V_CBOW_LAZY_MIN_BLOCKS_CH16=2
run cap3_3_2 -DGN2V_CBOW_LAZY_MIN_BLOCKS_CH4=3 -DGN2V_CBOW_LAZY_MIN_BLOCKS_CH8=3 -DGN2V_CBOW_LAZY_MIN_BLOCKS_CH16=2
python -c "from embiggen_amd import _lib; _lib.build(force=True)"
cat $L
