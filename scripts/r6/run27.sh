#!/bin/bash
# round 6, call 27: a rank of N with ONE extraction group a round (one scan of all ranks' walks) against two
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6; cd $R
show() { tail -1 $1 | python -c "import sys,json; l=json.loads(sys.stdin.readline()); r=l['roofline']; print(sys.argv[1], '%.4g'%l['value'], 'kernel %.4g'%r.get('kernel_pairs_per_s'), '%.2f ms x %d'%(r['avg_launch_ms'], r['launches']), l['config']['parallelism'][-75:], l.get('error',''))" $1; }
run() { tag=$1; shift; timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/r6/g_$tag.json 2> gpurun_out/r6/g_$tag.err; show gpurun_out/r6/g_$tag.json; }
run p8_g8 --phantom-world 8
run p8_g16 --phantom-world 8 --group-parts 16
run p4_g4 --phantom-world 4
run p4_g8 --phantom-world 4 --group-parts 8
run p8_g16_inline --phantom-world 8 --group-parts 16 --overlap off
