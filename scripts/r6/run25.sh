#!/bin/bash
# round 6, call 25: CUs left to the preparation (the training kernel on a CU-masked stream)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6; cd $R
show() { tail -1 $1 | python -c "import sys,json; l=json.loads(sys.stdin.readline()); r=l['roofline']; print(sys.argv[1], '%.4g'%l['value'], 'kernel %.4g'%r.get('kernel_pairs_per_s'), '%.2f ms x %d'%(r['avg_launch_ms'], r['launches']), l.get('error',''))" $1; }
run() { tag=$1; shift; timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/r6/z_$tag.json 2> gpurun_out/r6/z_$tag.err; show gpurun_out/r6/z_$tag.json; }
run r0
run r1 --reserve-cus 1
run r2 --reserve-cus 2
run r4 --reserve-cus 4
run p8_r0 --phantom-world 8
run p8_r1 --phantom-world 8 --reserve-cus 1
run p8_r2 --phantom-world 8 --reserve-cus 2
run p8_r4 --phantom-world 8 --reserve-cus 4
