#!/bin/bash
# same-box check: default CBOW bench with today's dispatch / launch-bound changes vs the build before them
# (ab_old_csrc/ = `git show 73e0f7c:embiggen_amd/csrc/{gn2v_api.hip,cbow_lazy_kernel.h}`, made for this run and removed after:
#  head 3.539 / 3.540e8 centres/s, before 3.541 / 3.541e8 -- no change at d = 128; the box was 7 % slower than the one
#  the committed r03_cbow profile was taken on)
mkdir -p gpurun_out
one() { timeout 900 python bench.py --model cbow --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); r=l['roofline']; print('$1 %.4e frac %.3f launch %.3f ms'%(l['value'], r['frac'], r['avg_launch_ms']))"; }
one head; one head
cp ab_old_csrc/gn2v_api.hip ab_old_csrc/cbow_lazy_kernel.h embiggen_amd/csrc/
python -c "from embiggen_amd import _lib; _lib.build(force=True)" > /dev/null 2>&1
one before; one before
