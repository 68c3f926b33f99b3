#!/bin/bash
# round 6, call 33: the final tree -- whole GPU suite, smoke, the driver's bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r6/final_suite.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6/final_suite.log
tail -3 gpurun_out/r6/final_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6/final_smoke.log 2>&1; tail -2 gpurun_out/r6/final_smoke.log
timeout 900 python bench.py > gpurun_out/r6/final_bench.json 2> gpurun_out/r6/final_bench.err; cut -c1-330 gpurun_out/r6/final_bench.json
