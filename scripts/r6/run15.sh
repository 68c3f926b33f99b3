#!/bin/bash
# round 6, fifteenth GPU call: cell_rows in closed form -- the alias tables must stay bit-equal to
# the oracle's (block tests, smoke), then the bench lines
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_resident.py tests/test_gpu_configs.py tests/test_gpu_world.py -q -x > gpurun_out/r6/t15.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r6/t15.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6/smoke15.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r6/smoke15.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r6/bench15.json 2> gpurun_out/r6/bench15.err
timeout 1500 python bench.py --no-cpu-baseline --nodes 100000000 --steps 16 --warmup 8 > gpurun_out/r6/bench15_100m.json 2> gpurun_out/r6/bench15_100m.err
tail -3 gpurun_out/r6/t15.log; tail -2 gpurun_out/r6/smoke15.log
for f in gpurun_out/r6/bench15.json gpurun_out/r6/bench15_100m.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"].get("frac"), d["roofline"].get("kernel_pairs_per_s"))
PY
done
