#!/bin/bash
# round 6, call 34: two lanes of round buffers in gn2v_train_blocks (round t + 1 prepared while round t trains)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6; cd $R
timeout 2000 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_configs.py tests/test_gpu_api.py tests/test_gpu_quality_gates.py tests/test_gpu_bench_contract.py tests/test_gpu_integration_doc.py -x -q -m gpu 2>&1 | tail -4
show() { tail -1 $1 | python -c "import sys,json; l=json.loads(sys.stdin.readline()); r=l['roofline']; print(sys.argv[1], '%.4g'%l['value'], 'kernel %.4g'%r.get('kernel_pairs_per_s'), 'frac %.3f'%r['frac'], '%.2f ms x %d'%(r['avg_launch_ms'], r['launches']), l['hbm_peak_gb']['device_in_use_now'], l.get('error',''))" $1; }
run() { tag=$1; shift; timeout 900 env "$@" python bench.py --no-cpu-baseline $EXTRA > gpurun_out/r6/q_$tag.json 2> gpurun_out/r6/q_$tag.err; show gpurun_out/r6/q_$tag.json; }
run lanes2 A=1
run lanes1 GN2V_ROUND_LANES=1
run lanes2_b A=1
run lanes1_b GN2V_ROUND_LANES=1
EXTRA="--nodes 2449029 --m 25" run c4_lanes2 A=1
EXTRA="--nodes 2449029 --m 25" run c4_lanes1 GN2V_ROUND_LANES=1
EXTRA="--nodes 169343 --m 7 --return-weight 2.0 --explore-weight 0.5 --walks 169343" run c3_lanes2 A=1
EXTRA="--nodes 169343 --m 7 --return-weight 2.0 --explore-weight 0.5 --walks 169343" run c3_lanes1 GN2V_ROUND_LANES=1
