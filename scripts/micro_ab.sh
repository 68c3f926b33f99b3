#!/bin/bash
# same-box micro A/B of the block kernel's launch parameters on the default bench
out=${1:-gpurun_out/micro_ab.log}; : > $out
bench() { python bench.py --steps 8 --warmup 4 --no-cpu-baseline "$@" 2>>$out | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])" | tee -a $out; }
bench --record 16
bench --record 32
bench --parts 64
bench --parts 16
bench --round-walks 8388608
bench --record 16
