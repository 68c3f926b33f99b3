#!/bin/bash
# round 6, call 21: a rank of 8 with its preparation in line under rocprofv3 (uncontended kernel times)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_p8i/stats -o stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --phantom-world 8 --overlap off > $R/gpurun_out/r6/p8i_prof.log 2>&1
cd $R
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_p8i/stats/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:22]: print(r["Name"][:130], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us", r["Percentage"])
PY
