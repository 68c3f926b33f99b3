#!/bin/bash
# PMC for the 100 M graph with the current code (whole rounds), + counter list
mkdir -p gpurun_out
( cd /tmp && export TMPDIR=/tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r3_counters.txt 2>&1 )
grep -i -c "name" gpurun_out/r3_counters.txt
bash scripts/profile_bench.sh r03_100m_a --nodes 100000000 --steps 8 --warmup 8 > gpurun_out/r3_prof_100m_a.log 2>&1
tail -5 gpurun_out/r3_prof_100m_a.log
