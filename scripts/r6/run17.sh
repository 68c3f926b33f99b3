#!/bin/bash
# round 6, seventeenth GPU call: episodes of a rank on two alternating streams (the next part's
# kernel fills the tail of the one before) -- a rank of 8 and of 2 without a fabric, A/B
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6; cd $R
show() { tail -1 $1 | python -c "import sys,json; l=json.loads(sys.stdin.readline()); r=l['roofline']; print(sys.argv[1], l['value'], l['ms_per_step'], 'kernel', r.get('kernel_pairs_per_s'), r['avg_launch_ms'], r['launches'])" $1; }
for w in 8 2; do
for es in 1 2; do
  GN2V_EPISODE_STREAMS=$es timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --phantom-world $w > gpurun_out/r6/ep_streams_${es}_w$w.json 2> gpurun_out/r6/ep_streams_${es}_w$w.err
  show gpurun_out/r6/ep_streams_${es}_w$w.json
done; done
timeout 1500 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_bench_contract.py tests/test_gpu_world.py -x -q -m gpu 2>&1 | tail -5
