#!/bin/bash
# experiment: which band of row "hotness" (share of the cell's edge endpoints) should be updated
# with atomics?  speed on the bench graph + quality on BA 1 M for several bands
out=${1:-gpurun_out/hub_band.log}; : > $out
for band in "0 0" "40 0" "12 0" "12 9" "12 7" "10 7" "9 6" "14 9"; do
  set -- $band
  echo "### lo=$1 hi=$2" | tee -a $out
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --hot-band $1:$2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['roofline']['frac'])" | tee -a $out
  python scripts/quality_probe.py --epochs 3 --modes blocks:16:8 --hot-band $1:$2 2>&1 | grep blocks | cut -c1-175 | tee -a $out
done
