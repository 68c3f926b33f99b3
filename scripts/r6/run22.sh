#!/bin/bash
# round 6, call 22: extraction by own centre where a rank owns fewer centres than the group holds
# contexts; phantom rank generates only its own walks inside the timed region
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6; cd $R
timeout 1500 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_world.py tests/test_gpu_bench_contract.py -x -q -m gpu 2>&1 | tail -4
show() { tail -1 $1 | python -c "import sys,json; l=json.loads(sys.stdin.readline()); r=l['roofline']; print(sys.argv[1], '%.4g'%l['value'], 'kernel %.4g'%r.get('kernel_pairs_per_s'), '%.2f ms x %d'%(r['avg_launch_ms'], r['launches']), l['config']['parallelism'][-60:])" $1; }
run() { tag=$1; shift; timeout 900 env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline $EXTRA > gpurun_out/r6/x_$tag.json 2> gpurun_out/r6/x_$tag.err; show gpurun_out/r6/x_$tag.json; }
EXTRA="--phantom-world 8" run p8 A=1
EXTRA="--phantom-world 8 --overlap off" run p8_inline A=1
EXTRA="--phantom-world 4" run p4 A=1
EXTRA="--phantom-world 2" run p2 A=1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_p8x/stats -o stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --phantom-world 8 --overlap off > $R/gpurun_out/r6/p8x_prof.log 2>&1
cd $R
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_p8x/stats/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:10]: print(r["Name"][:100], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us", r["Percentage"])
PY
