#!/bin/bash
# round 6, call 28: the round plan made for the round that is trained (cap) -- several ranks hold a round in one group
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6; cd $R
timeout 2000 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_world.py tests/test_gpu_bench_contract.py tests/test_gpu_configs.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -4
show() { tail -1 $1 | python -c "import sys,json; l=json.loads(sys.stdin.readline()); r=l['roofline']; print(sys.argv[1], '%.4g'%l['value'], 'kernel %.4g'%r.get('kernel_pairs_per_s'), '%.2f ms x %d'%(r['avg_launch_ms'], r['launches']), l['config']['parallelism'][-75:], l.get('error',''))" $1; }
run() { tag=$1; shift; timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/r6/h_$tag.json 2> gpurun_out/r6/h_$tag.err; show gpurun_out/r6/h_$tag.json; }
run one
run p8 --phantom-world 8
run p4 --phantom-world 4
run p2 --phantom-world 2
run p8_100m --phantom-world 8 --nodes 100000000 --steps 8 --warmup 4
