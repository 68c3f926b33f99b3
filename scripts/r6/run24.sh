#!/bin/bash
# round 6, call 24: one GPU with the preparation in line under rocprofv3 (uncontended kernel times)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_1i/stats -o stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --entry python --overlap off > $R/gpurun_out/r6/1i_prof.log 2>&1
cd $R
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_1i/stats/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:16]: print(r["Name"][:150], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us", r["Percentage"])
PY
