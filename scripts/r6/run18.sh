#!/bin/bash
# round 6, eighteenth GPU call: what slows the training kernel of a rank of 8 (1.87e9 against
# 2.2e9 pairs/s on one GPU)?  preparation in line, longer rounds, index order
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6; cd $R
show() { tail -1 $1 | python -c "import sys,json; l=json.loads(sys.stdin.readline()); r=l['roofline']; print(sys.argv[1], '%.4g'%l['value'], 'kernel %.4g'%r.get('kernel_pairs_per_s'), '%.2f ms x %d'%(r['avg_launch_ms'], r['launches']), l['config']['parallelism'][-120:])" $1; }
run() { tag=$1; shift; timeout 900 env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --phantom-world 8 $EXTRA > gpurun_out/r6/p8_$tag.json 2> gpurun_out/r6/p8_$tag.err; show gpurun_out/r6/p8_$tag.json; }
run base A=1
EXTRA="--overlap off" run inline A=1
run rounds21 GN2V_ROUND_MIN_WALKS=2097152
run nolpt GN2V_RESIDENT_LPT=0
EXTRA="--overlap off" run inline_rounds21 GN2V_ROUND_MIN_WALKS=2097152
