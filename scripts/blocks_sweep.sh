#!/bin/bash
# sweep of the block trainer's plan on one GPU (parts x record x slices); prints one line per run
out=${1:-gpurun_out/blocks_sweep.log}
: > $out
run() { echo "### $*" >> $out; python bench.py --steps 3 --warmup 1 --no-cpu-baseline --parallelism blocks "$@" 2>>$out | tail -1 >> $out; }
for parts in 4 8 16 32; do for record in 16 32; do run --parts $parts --slices 8 --record $record; done; done
run --parts 16 --slices 4 --record 32
run --parts 16 --slices 16 --record 32
run --parts 32 --slices 8 --record 24
python - "$out" <<'PY'
import json,sys
tag=None
for l in open(sys.argv[1]):
    if l.startswith("###"): tag=l[4:].strip()
    elif l.startswith("{"):
        d=json.loads(l); r=d["roofline"]
        print(f"{tag:45s} value {d['value']:.3e}  ms/step {d['ms_per_step']:8.1f}  kernel frac {r['frac']:.3f}  launch ms {r['avg_launch_ms']:.1f}")
PY
