#!/bin/bash
# round 6, call 35: the final tree (two lanes while the second set of round buffers is small) -- suite, smoke, bench lines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r6/final_suite.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6/final_suite.log
tail -3 gpurun_out/r6/final_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6/final_smoke.log 2>&1; tail -1 gpurun_out/r6/final_smoke.log
timeout 900 python bench.py > gpurun_out/r6/final_bench.json 2> gpurun_out/r6/final_bench.err
show() { tail -1 $1 | python -c "import sys,json; l=json.loads(sys.stdin.readline()); r=l['roofline']; print(sys.argv[1], '%.4g'%l['value'], 'kernel %.4g'%r.get('kernel_pairs_per_s'), 'frac %.3f'%r['frac'], '%.2f ms x %d'%(r['avg_launch_ms'], r['launches']), l['hbm_peak_gb']['device_in_use_now'], l.get('error',''))" $1; }
show gpurun_out/r6/final_bench.json
run() { tag=$1; shift; timeout 900 python bench.py --no-cpu-baseline "$@" > gpurun_out/r6/s_$tag.json 2> gpurun_out/r6/s_$tag.err; show gpurun_out/r6/s_$tag.json; }
run c4 --nodes 2449029 --m 25
run c3 --nodes 169343 --m 7 --return-weight 2.0 --explore-weight 0.5 --walks 169343
run m1 --nodes 1000000
timeout 900 python scripts/small_fits.py 2>&1 | grep "^BA" | tail -6
