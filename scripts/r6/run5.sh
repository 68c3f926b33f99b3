#!/bin/bash
# round 6, fifth GPU call: does folding same-centre gradients before the atomics pay?  bench A/B by
# sort key (top 8 bits of the centre / the whole centre) and round length
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
run() { # name, env..., -- args
  name=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py --no-cpu-baseline "$@" > gpurun_out/r6/fold_$name.json 2> gpurun_out/r6/fold_$name.err
  python - gpurun_out/r6/fold_$name.json "$name" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(f"{sys.argv[2]:28s} value {d['value']:.4g}  kernel {r.get('kernel_pairs_per_s',0):.4g}  pairs/hand-over {r['mean_centre_run']:.3f}  launch_ms {r['avg_launch_ms']:.1f} x{r['launches']}  rounds of {d['config']['parallelism'].split('rounds of ')[1][:40]}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run sort8_default GN2V_X=0 --
run sort32_default GN2V_SORT_CENTRE_BITS=32 --
run sort32_round8m GN2V_SORT_CENTRE_BITS=32 -- --round-walks 8388608
run sort8_round8m GN2V_X=0 -- --round-walks 8388608
run sort32_r64 GN2V_SORT_CENTRE_BITS=32 GN2V_ROUNDS_PER_EPOCH=64 GN2V_ROUND_MIN_WALKS=16384 --
run sort8_r64 GN2V_ROUNDS_PER_EPOCH=64 GN2V_ROUND_MIN_WALKS=16384 --
run sort8_r32 GN2V_ROUNDS_PER_EPOCH=32 GN2V_ROUND_MIN_WALKS=16384 --
