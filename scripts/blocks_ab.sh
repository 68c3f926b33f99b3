#!/bin/bash
# A/B of the one-GPU bench lines: walk-ordered kernel vs the block trainer in several plans.
# usage: bash scripts/blocks_ab.sh <out-file>  (on the GPU box, through gpurun)
out=${1:-gpurun_out/blocks_ab.log}
: > $out
run() { echo "### $*" >> $out; python bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" 2>>$out | tail -1 >> $out; }
run --parallelism single
run --parallelism blocks
run --parallelism blocks --overlap on
run --parallelism blocks --parts 16
run --parallelism blocks --parts 16 --slices 8
run --parallelism blocks --slices 8
run --parallelism blocks --record 32
run --parallelism blocks --record 8
python - <<'PY' >> $out
import json,sys
for l in open(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/blocks_ab.log"):
    pass
PY
python - "$out" <<'PY'
import json,sys
tag=None
for l in open(sys.argv[1]):
    if l.startswith("###"): tag=l[4:].strip()
    elif l.startswith("{"):
        d=json.loads(l); r=d["roofline"]
        print(f"{tag:55s} value {d['value']:.3e}  ms/step {d['ms_per_step']:8.1f}  kernel frac {r['frac']:.3f}  launch ms {r['avg_launch_ms']:.1f}")
PY
