#!/bin/bash
# round 6, call 31: one GPU -- fewer, larger extraction groups a round (6 groups of 30 parts is the plan's choice)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6; cd $R
show() { tail -1 $1 | python -c "import sys,json; l=json.loads(sys.stdin.readline()); r=l['roofline']; print(sys.argv[1], '%.4g'%l['value'], 'kernel %.4g'%r.get('kernel_pairs_per_s'), '%.2f ms x %d'%(r['avg_launch_ms'], r['launches']), l['config']['parallelism'][-95:], l.get('error',''))" $1; }
run() { tag=$1; shift; timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/r6/k_$tag.json 2> gpurun_out/r6/k_$tag.err; show gpurun_out/r6/k_$tag.json; }
run c_entry
run py_g30 --entry python --group-parts 30
run py_g60 --entry python --group-parts 60
run py_g89 --entry python --group-parts 89
run py_g178 --entry python --group-parts 178
