#!/bin/bash
# round 6, call 32: one extraction group a round on one GPU too -- suites of the block path, then the bench lines
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6; cd $R
timeout 2400 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_resident.py tests/test_gpu_configs.py tests/test_gpu_quality_gates.py tests/test_gpu_api.py tests/test_gpu_bench_contract.py tests/test_gpu_default_vs_oracle.py -x -q -m gpu 2>&1 | tail -4
show() { tail -1 $1 | python -c "import sys,json; l=json.loads(sys.stdin.readline()); r=l['roofline']; print(sys.argv[1], '%.4g'%l['value'], 'kernel %.4g'%r.get('kernel_pairs_per_s'), 'frac %.3f'%r['frac'], '%.2f ms x %d'%(r['avg_launch_ms'], r['launches']), 'first_fit', l.get('first_fit_s'), l['config']['parallelism'][-80:], l.get('error',''))" $1; }
run() { tag=$1; shift; timeout 900 python bench.py --no-cpu-baseline "$@" > gpurun_out/r6/m_$tag.json 2> gpurun_out/r6/m_$tag.err; show gpurun_out/r6/m_$tag.json; }
run one
run one_again
run c4 --nodes 2449029 --m 25
run c3 --nodes 169343 --m 7 --return-weight 2.0 --explore-weight 0.5 --walks 169343
run m1 --nodes 1000000
run b100m --nodes 100000000 --steps 8 --warmup 4
run p8 --phantom-world 8
