#!/bin/bash
# round 6, sixteenth GPU call: where does a small graph's time go?  config 3's shape under rocprofv3
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6
ARGS="--nodes 169343 --m 7 --return-weight 2.0 --explore-weight 0.5 --walks 169343 --steps 30 --warmup 10 --no-cpu-baseline"
python3 $R/bench.py $ARGS > $R/gpurun_out/r6/c3_plain.json 2> $R/gpurun_out/r6/c3_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3/stats -o stats -- python3 $R/bench.py $ARGS > $R/gpurun_out/r6/c3_prof.log 2>&1
cd $R
python - <<'PY'
import json,csv,glob
d=json.load(open("gpurun_out/r6/c3_plain.json")); r=d["roofline"]
print("value", d["value"], "kernel", r.get("kernel_pairs_per_s"), "launch_ms", r["avg_launch_ms"], "x", r["launches"], d["config"]["parallelism"][-130:])
f=glob.glob("gpurun_out/prof_c3/stats/**/*kernel_stats.csv", recursive=True)[0]
tot=0
rows=list(csv.DictReader(open(f)))
for r in rows[:16]: print(r["Name"][:80], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us", r["Percentage"])
PY
