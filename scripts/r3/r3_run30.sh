#!/bin/bash
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r3_bench30_default.json 2> gpurun_out/r3_bench30_default.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r3_bench30_default.json") if l.startswith("{")][-1]); r=d["roofline"]
print("value %.4e ms/step %.1f frac_hbm %.3f frac %.3f"%(d["value"], d["ms_per_step"], r["frac_hbm"], r["frac"]))
print(d["cpu_baseline"])
PY
timeout 600 python bench.py --model cbow --cpu-seconds 10 > gpurun_out/r3_bench30_cbow.json 2> gpurun_out/r3_bench30_cbow.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r3_bench30_cbow.json") if l.startswith("{")][-1]); r=d["roofline"]
print("cbow value %.4e frac_hbm %s frac %.3f"%(d["value"], r["frac_hbm"], r["frac"]))
print(d["cpu_baseline"])
PY
timeout 900 python -m pytest tests/test_gpu_bench_contract.py -m gpu -q 2>&1 | tail -2
