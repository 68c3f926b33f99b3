#!/bin/bash
# Instruction mix and issue activity of the resident-cell kernel on the bench graph (run through
# gpurun from the repo root): one rocprofv3 --pmc pass per counter group (with --kernel-trace for
# the kernel names and durations; no other trace domain), digest ->
# gpurun_out/resident_counters.json (per-launch means + pairs per launch of the bench line)
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/resident_counters
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
ARGS="--steps 8 --warmup 0 --no-cpu-baseline"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT"; do
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/g$i" -o c -- python3 "$ROOTDIR/bench.py" $ARGS > "$OUT/g$i.log" 2>&1
  i=$((i+1))
done
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sgns_resident" in r["Kernel_Name"]:
            agg[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
res = {}
for name, per in agg.items():
    vals = [sum(v) for v in per.values()]
    res[name] = {"launches": len(vals), "mean_per_launch": sum(vals) / len(vals)}
# the bench line of the first pass: pairs per launch, for per-pair figures
for line in open(out + "/g0.log"):
    if line.startswith("{"):
        b = json.loads(line)
        pairs = b["value"] * b["ms_per_step"] * 1e-3 * b["steps"]
        res["bench"] = {"argv": b.get("argv"), "pairs": pairs, "launches": b["roofline"]["launches"],
                        "pairs_per_launch": pairs / b["roofline"]["launches"],
                        "kernel": b["roofline"]["kernel"]}
if "bench" in res:
    ppl = res["bench"]["pairs_per_launch"]
    res["per_pair"] = {k: v["mean_per_launch"] / ppl for k, v in res.items()
                       if isinstance(v, dict) and "mean_per_launch" in v}
json.dump(res, open(out + "/../resident_counters.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
