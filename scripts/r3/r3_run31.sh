#!/bin/bash
# the N = 8 bench line end to end on one GPU (8 processes share it, gloo as the fabric): host logic only
mkdir -p gpurun_out
timeout 1500 python bench.py --gpus 8 --backend gloo --share-device --nodes 2000000 --walks 32768 --steps 4 --warmup 2 > gpurun_out/r3_bench31_gloo8.json 2> gpurun_out/r3_bench31_gloo8.err
echo "rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r3_bench31_gloo8.json") if l.startswith("{")][-1])
print(d["n_gpus"], "%.3e"%d["value"], d["finite"], d["config"]["parallelism"])
print(json.dumps(d["distributed"])[:1500])
PY
tail -5 gpurun_out/r3_bench31_gloo8.err
timeout 1500 python -m pytest tests/test_gpu_configs.py -m gpu -q -k "node_order" 2>&1 | tail -2
