#!/bin/bash
# bash scripts/pmc_sweep.sh : WRITE_SIZE / FETCH_SIZE per launch of the SGNS kernel for a few (k, mode) settings
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/pmc_sweep
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
i=0
for cfg in "--k 10 --mode write_through" "--k 10 --mode write_back" "--k 5 --mode write_through" "--k 0 --mode write_through" "--k 10 --window 2 --mode write_through"; do
  for ctr in WRITE_SIZE FETCH_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$OUT/c${i}_$ctr" -o p -- python3 "$ROOTDIR/scripts/pmc_probe.py" $cfg > "$OUT/c${i}_$ctr.log" 2>&1
    python3 - "$OUT/c${i}_$ctr" "$ctr" "$OUT/c${i}_$ctr.log" <<'PY'
import csv, glob, json, sys
d, ctr, log = sys.argv[1:4]
info = json.loads([l for l in open(log) if l.startswith("{")][-1])
vals = [float(r["Counter_Value"]) for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        for r in csv.DictReader(open(f)) if "sgns_kernel" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
kib = sum(vals) / len(vals)
print(info, ctr, "KiB/launch", kib, "bytes/pair", kib * 1024 / info["pairs_per_launch"], "bytes/centre", kib * 1024 / info["centres_per_launch"])
PY
  done
  i=$((i+1))
done
