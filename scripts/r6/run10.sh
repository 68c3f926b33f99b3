#!/bin/bash
# round 6, tenth GPU call: counted scatter against the radix sort on ONE box
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline "$@" > gpurun_out/r6/ab_$name.json 2> gpurun_out/r6/ab_$name.err
  python - gpurun_out/r6/ab_$name.json "$name" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(f"{sys.argv[2]:26s} value {d['value']:.4g}  kernel {r.get('kernel_pairs_per_s',0):.4g}  frac {r.get('frac',0):.3f}  launch_ms {r['avg_launch_ms']:.1f} x{r['launches']}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run scatter_1 GN2V_SCATTER=1 --
run sort_1 GN2V_SCATTER=0 --
run scatter_2 GN2V_SCATTER=1 --
run sort_2 GN2V_SCATTER=0 --
run scatter_phantom8 GN2V_SCATTER=1 -- --phantom-world 8
run sort_phantom8 GN2V_SCATTER=0 -- --phantom-world 8
run scatter_100m GN2V_SCATTER=1 -- --nodes 100000000 --steps 16 --warmup 8
run sort_100m GN2V_SCATTER=0 -- --nodes 100000000 --steps 16 --warmup 8
