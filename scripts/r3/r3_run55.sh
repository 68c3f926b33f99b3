#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q > gpurun_out/r3_gputests55.log 2>&1
tail -6 gpurun_out/r3_gputests55.log
for d in 128 160 200 256 512; do
  timeout 600 python bench.py --model cbow --d $d --nodes 1000000 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); r=l['roofline']; print('d=$d %.3e centres/s frac %.3f %s launch %.2f ms finite %s'%(l['value'], r['frac'], r['kernel'], r['avg_launch_ms'], l['finite']))"
done
timeout 600 python bench.py --model cbow --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); r=l['roofline']; print('default cbow bench %.3e frac %.3f %s'%(l['value'], r['frac'], r['kernel']))"
