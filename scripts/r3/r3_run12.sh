#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_fuzz.py tests/test_gpu_api.py tests/test_gpu_bench_contract.py -m gpu -q -x > gpurun_out/r3_gputests12.log 2>&1
tail -15 gpurun_out/r3_gputests12.log
for lazy in 1 0 1 0; do
  GN2V_CBOW_LAZY=$lazy timeout 600 python bench.py --model cbow --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3_bench12_cbow_lazy$lazy.json 2> gpurun_out/r3_bench12_cbow_lazy$lazy.err
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r3_bench12_cbow_lazy$lazy.json") if l.startswith("{")][-1]); r=d["roofline"]
print("lazy=$lazy value %.3e frac %.3f launch %.2f ms finite %s"%(d["value"], r["frac"], r["avg_launch_ms"], d["finite"]))
PY
done
