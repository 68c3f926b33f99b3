#!/bin/bash
# round 6, first GPU call: the whole -m gpu suite after the redraw of cell-local negatives, the
# link-quality A/B at config 3's shape (round length), one bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r6/t1.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r6/t1.log
Q="python scripts/quality_probe.py --nodes 169343 --m 7 --walks 1693430 --epochs 3"
( $Q --round-walks 1693430 --modes blocks:4:256,blocks:1:8,atomic
  $Q --round-walks 423358 --modes blocks:4:256
  $Q --round-walks 211679 --modes blocks:4:256
  $Q --round-walks 105840 --modes blocks:4:256 ) > gpurun_out/r6/q1.log 2>&1
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r6/bench1.json 2> gpurun_out/r6/bench1.err
tail -3 gpurun_out/r6/t1.log; cat gpurun_out/r6/q1.log | grep -v "^\[" ; cat gpurun_out/r6/bench1.json | cut -c1-600
