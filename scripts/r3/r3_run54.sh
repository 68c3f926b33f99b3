#!/bin/bash
# lazy CBOW window at d = 160 / 200 / 256 (CH = 4, 64 KB LDS budget): register cap 2 / 3 / 4 blocks per CU
mkdir -p gpurun_out
L=gpurun_out/r3_cbow_wide_cap.log; : > $L
for cap in 3 2 4; do
  GN2V_HIPCC_FLAGS="-DGN2V_CBOW_LAZY_MIN_BLOCKS_CH4=$cap" python -c "from embiggen_amd import _lib; _lib.build(force=True)" || exit 1
  for d in 160 200 256; do
    timeout 600 python bench.py --model cbow --d $d --nodes 1000000 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r3_cbow_cap.json 2>/dev/null
    python - "$cap" "$d" >> $L <<'PY'
import json, sys
cap, d = sys.argv[1:3]
l = json.loads([x for x in open("gpurun_out/r3_cbow_cap.json") if x.startswith("{")][-1]); r = l["roofline"]
print(f"CH4 cap {cap} d={d:5s} {l['value']:.3e} centres/s frac {r['frac']:.3f} launch {r['avg_launch_ms']:.2f} ms finite {l['finite']}")
PY
  done
done
GN2V_CBOW_LAZY_LDS_KB=40 timeout 600 python bench.py --model cbow --d 160 --nodes 1000000 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); print('plain d=160 %.3e frac %.3f'%(l['value'], l['roofline']['frac']))" >> $L
python -c "from embiggen_amd import _lib; _lib.build(force=True)"
timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -q -k "cbow or cache" 2>&1 | tail -3 >> $L
cat $L
