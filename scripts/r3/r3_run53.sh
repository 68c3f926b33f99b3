#!/bin/bash
# wide rows: plain CBOW kernel (40 KB LDS budget) vs the window-cache kernels with 64 KB per workgroup
mkdir -p gpurun_out
L=gpurun_out/r3_cbow_wide_lds.log; : > $L
for kb in 40 64; do for lazy in 1 0; do for d in 200 256; do
  GN2V_CTX_CACHE_LDS_KB=$kb GN2V_CBOW_LAZY=$lazy timeout 600 python bench.py --model cbow --d $d --nodes 1000000 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r3_cbow_lds.json 2>/dev/null
  python - "$kb" "$lazy" "$d" >> $L <<'PY'
import json, sys
kb, lazy, d = sys.argv[1:4]
l = json.loads([x for x in open("gpurun_out/r3_cbow_lds.json") if x.startswith("{")][-1]); r = l["roofline"]
print(f"lds {kb} KB lazy {lazy} d={d:5s} {l['value']:.3e} centres/s frac {r['frac']:.3f} launch {r['avg_launch_ms']:.2f} ms finite {l['finite']}")
PY
done; done; done
cat $L
