#!/bin/bash
# round 6, eleventh GPU call: how many bits of the centre should the sort key of resident plans
# keep?  (the band a record's central rows fall into: 8 bits = 1/256 of the table)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline "$@" > gpurun_out/r6/bits_$name.json 2> gpurun_out/r6/bits_$name.err
  python - gpurun_out/r6/bits_$name.json "$name" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(f"{sys.argv[2]:26s} value {d['value']:.4g}  kernel {r.get('kernel_pairs_per_s',0):.4g}  frac {r.get('frac',0):.3f}  launch_ms {r['avg_launch_ms']:.1f} x{r['launches']}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for b in 8 4 12 16 8; do run bench_bits$b GN2V_SORT_CENTRE_BITS=$b --; done
for b in 8 12 16; do run 100m_bits$b GN2V_SORT_CENTRE_BITS=$b -- --nodes 100000000 --steps 16 --warmup 8; done
for b in 8 12; do run ph8_bits$b GN2V_SORT_CENTRE_BITS=$b -- --phantom-world 8; done
