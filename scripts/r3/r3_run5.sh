#!/bin/bash
# (a) TLB / L2 counters of the training kernel on the 10 M and the 100 M graph, (b) chunked visiting order
mkdir -p gpurun_out/r3_diag
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for nodes in 10000000 100000000; do
  for ctr in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"; do
    tag=$(echo $ctr | cut -d' ' -f1)
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/r3_diag/${nodes}_$tag -o c -- python3 $R/bench.py --nodes $nodes --steps 4 --warmup 0 --no-cpu-baseline --round-walks 4194304 > $R/gpurun_out/r3_diag/${nodes}_$tag.log 2>&1
  done
done
cd $R
python - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob("gpurun_out/r3_diag/*/*counter_collection.csv")):
    agg=collections.defaultdict(float); n=collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if "sgns_block_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    print(f.split("/")[-2], {k:(v, n[k]) for k,v in agg.items()})
PY
for chunk in 0 64 256; do
  GN2V_BLOCK_CHUNK=$chunk timeout 900 python bench.py --nodes 100000000 --steps 8 --warmup 8 --no-cpu-baseline --record 32 > gpurun_out/r3_bench5_100m_chunk$chunk.json 2> gpurun_out/r3_bench5_100m_chunk$chunk.err
done
for chunk in 64 256; do
  GN2V_BLOCK_CHUNK=$chunk timeout 600 python bench.py --steps 8 --warmup 8 --no-cpu-baseline --record 32 > gpurun_out/r3_bench5_chunk$chunk.json 2> gpurun_out/r3_bench5_chunk$chunk.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_bench5*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); r=d["roofline"]
        print(f, "value %.3e"%d["value"], "frac %.3f"%r["frac"], "sched %.3f"%r["frac_scheduled"], "run %.2f"%r["mean_centre_run"], "launch %.1f ms"%r["avg_launch_ms"])
    except Exception as e: print(f, "failed", e)
PY
