#!/bin/bash
# embedding sizes: are there cliffs away from d = 128?
mkdir -p gpurun_out
L=gpurun_out/r3_d_sweep.log; : > $L
for model in skipgram cbow; do
for d in 8 16 32 64 100 128 200 256 512 1024; do
  timeout 600 python bench.py --model $model --d $d --nodes 1000000 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r3_d_$model_$d.json 2>/dev/null
  python - "$model" "$d" >> $L <<'PY'
import json, sys
m, d = sys.argv[1], sys.argv[2]
try:
    l = json.loads([x for x in open(f"gpurun_out/r3_d_{d}.json") if x.startswith("{")][-1]); r = l["roofline"]
    print(f"{m:9s} d={d:5s} value {l['value']:.3e} {l['unit']} frac {r['frac']:.3f} kernel {r['kernel']} launch {r['avg_launch_ms']:.2f} ms finite {l['finite']}")
except Exception as e:
    print(m, d, "FAILED", e)
PY
done; done
cat $L
