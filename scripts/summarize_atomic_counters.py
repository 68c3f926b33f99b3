"""Digest of scripts/atomic_counters.sh: per kernel (the atomic probe's four shapes + its store
mode, at two table sizes; the bench's resident kernel) the TCC counters per launch next to the
launch's duration from the same pass's kernel trace, and the figures derived from them:

    atomic_sectors_per_channel_cycle = TCC_ATOMIC_SECTORS / TCC_CYCLE   (both summed over channels)
    tcc_busy                         = TCC_BUSY / TCC_CYCLE
    tag_stall                        = TCC_TAG_STALL / TCC_CYCLE
    dword_atomics_per_s              = TCC_ATOMIC_SECTORS x 8 / duration  (a 32 B sector = 8 f32 adds)

A kernel at the ceiling of the L2 atomic units shows the same sectors per channel-cycle as the
probe, whatever else it does."""
import collections
import csv
import glob
import json
import os
import sys


def passes(out, prefix):
    """{kernel name: {counter: [per-dispatch sums]}, '_ns': {kernel: [durations]}} of one program."""
    counters = collections.defaultdict(lambda: collections.defaultdict(dict))
    dur = collections.defaultdict(dict)
    for d in sorted(glob.glob(os.path.join(out, prefix + "_g*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                per = counters[r["Kernel_Name"]][r["Counter_Name"]]
                per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                dur[r["Kernel_Name"]].setdefault(os.path.basename(d), []).append(
                    int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return counters, dur


def digest(counters, dur, keep, skip_first=0):
    res = {}
    for name, per_counter in counters.items():
        if not keep(name):
            continue
        rec = {}
        for cname, per in per_counter.items():
            vals = [v for _, v in sorted(per.items(), key=lambda kv: int(kv[0]))][skip_first:]
            if vals:
                rec[cname] = sum(vals) / len(vals)
        # mean duration over all passes of the launches kept
        ns = [x for runs in dur.get(name, {}).values() for x in runs[skip_first:]]
        if ns:
            rec["launch_ms"] = sum(ns) / len(ns) / 1e6
            rec["launches_per_pass"] = len(ns) / max(len(dur[name]), 1)
        cyc = rec.get("TCC_CYCLE_sum")
        if cyc:
            for key, c in (("atomic_sectors_per_channel_cycle", "TCC_ATOMIC_SECTORS_sum"),
                           ("atomic_requests_per_channel_cycle", "TCC_ATOMIC_sum"),
                           ("tcc_busy", "TCC_BUSY_sum"), ("tag_stall", "TCC_TAG_STALL_sum"),
                           ("requests_per_channel_cycle", "TCC_REQ_sum")):
                if c in rec:
                    rec[key] = rec[c] / cyc
        if "TCC_ATOMIC_SECTORS_sum" in rec and "launch_ms" in rec:
            rec["dword_atomics_per_s"] = rec["TCC_ATOMIC_SECTORS_sum"] * 8 / (rec["launch_ms"] * 1e-3)
        if rec.get("TCC_EA0_ATOMIC_sum"):
            rec["ea_atomic_latency_cycles"] = (rec.get("TCC_EA0_ATOMIC_LEVEL_sum", 0.0)
                                               / rec["TCC_EA0_ATOMIC_sum"])
        res[name.split("(")[0][:80]] = rec
    return res


if __name__ == "__main__":
    out = sys.argv[1]
    res = {}
    for rows in (100000, 10000000):
        c, d = passes(out, f"probe_{rows}")
        # the probe launches every mode twice: a short warm-up, then the timed launch
        res[f"atomic_probe_{rows}_rows"] = digest(c, d, lambda n: "probe<" in n, skip_first=1)
    c, d = passes(out, "bench")
    res["bench"] = digest(c, d, lambda n: "sgns_resident" in n)
    for line in open(os.path.join(out, "bench_g0.log")):
        if line.startswith("{"):
            b = json.loads(line)
            res["bench_line"] = {"argv": b.get("argv"), "value": b["value"],
                                 "kernel": b["roofline"]["kernel"],
                                 "kernel_pairs_per_s": b["roofline"].get("kernel_pairs_per_s"),
                                 "frac": b["roofline"].get("frac"),
                                 "workload": b["config"]["workload"]}
    print(json.dumps(res, indent=1))
