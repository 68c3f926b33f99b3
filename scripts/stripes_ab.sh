#!/bin/bash
# speed and link quality of the optional centre stripes, same box
out=${1:-gpurun_out/stripes_ab.log}; : > $out
for v in 1 2 4 8; do
  python bench.py --steps 16 --warmup 8 --no-cpu-baseline --stripes $v 2>/dev/null | tail -1 |
    python -c "import sys,json; l=json.loads(sys.stdin.readline()); print('bench --steps 16 --stripes $v:', l['value'], 'pairs/s', l['ms_per_step'], 'ms/step, kernel', l['roofline']['frac'], l['config']['parallelism'][-75:])" | tee -a $out
done
q() { python scripts/quality_probe.py "$@" 2>&1 | tail -1 | sed "s|^|quality_probe $*: |" | tee -a $out; }
q --nodes 10000000 --walks 33554432 --modes blocks:32:8 --round-walks 8388608 --stripes 1
q --nodes 10000000 --walks 33554432 --modes blocks:32:8 --round-walks 8388608 --stripes 8
q --nodes 10000000 --walks 33554432 --modes blocks:32:8 --round-walks 67108864 --stripes 8
q --nodes 1000000 --m 7 --walks 10000000 --epochs 5 --modes blocks:2:8 --round-walks 10000000 --stripes 1
q --nodes 1000000 --m 7 --walks 10000000 --epochs 5 --modes blocks:2:8 --round-walks 10000000 --stripes 8
