#!/bin/bash
# same-box A/B of the block kernel's occupancy: default build vs -DGN2V_BLOCK_MIN_BLOCKS=<n>
out=${1:-gpurun_out/occupancy_ab.log}; : > $out
bench() { python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>>$out | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])" | tee -a $out; }
bench default
for n in 6 8; do
  GN2V_HIPCC_FLAGS="-DGN2V_BLOCK_MIN_BLOCKS=$n" python -c "from embiggen_amd import _lib; _lib.build(force=True)" >> $out 2>&1
  bench min_blocks_$n
done
python -c "from embiggen_amd import _lib; _lib.build(force=True)" >> $out 2>&1
bench default_again
