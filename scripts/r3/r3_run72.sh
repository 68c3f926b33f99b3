#!/bin/bash
# SkipGram block kernel at d = 200 / 256 / 512 (CH = 4 / 8): natural register use (3 / 2 waves per SIMD) vs caps
mkdir -p gpurun_out
L=gpurun_out/r3_block_wide_cap.log; : > $L
for cap in 1 4 3; do
  GN2V_HIPCC_FLAGS="-DGN2V_BLOCK_MIN_BLOCKS=$cap" python -c "from embiggen_amd import _lib; _lib.build(force=True)" > /dev/null 2>&1 || exit 1
  for d in 200 256 512; do
    timeout 600 python bench.py --d $d --nodes 1000000 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); r=l['roofline']; print('min blocks $cap d=$d %.3e pairs/s frac %.3f launch %.1f ms finite %s'%(l['value'], r['frac'], r['avg_launch_ms'], l['finite']))" >> $L
  done
done
python -c "from embiggen_amd import _lib; _lib.build(force=True)" > /dev/null 2>&1
cat $L
