#!/bin/bash
# round 6, nineteenth GPU call: a rank of 8 without a fabric under rocprofv3 --kernel-trace --stats
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_p8/stats -o stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --phantom-world 8 > $R/gpurun_out/r6/p8_prof.log 2>&1
cd $R
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_p8/stats/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:22]: print(r["Name"][:110], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us", r["Percentage"])
PY
tail -1 gpurun_out/r6/p8_prof.log | cut -c1-300
