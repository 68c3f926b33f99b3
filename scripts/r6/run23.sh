#!/bin/bash
# round 6, call 23: one GPU -- the training kernel alone (preparation in line) against the overlapped default
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6; cd $R
show() { tail -1 $1 | python -c "import sys,json; l=json.loads(sys.stdin.readline()); r=l['roofline']; print(sys.argv[1], '%.4g'%l['value'], 'kernel %.4g'%r.get('kernel_pairs_per_s'), '%.2f ms x %d'%(r['avg_launch_ms'], r['launches']), l['config']['parallelism'][-70:])" $1; }
run() { tag=$1; shift; timeout 900 env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline $EXTRA > gpurun_out/r6/y_$tag.json 2> gpurun_out/r6/y_$tag.err; show gpurun_out/r6/y_$tag.json; }
EXTRA="" run c_entry A=1
EXTRA="--entry python --overlap off" run py_inline A=1
EXTRA="--entry python --overlap on" run py_overlap A=1
