"""Turn the raw rocprofv3 CSVs of scripts/profile_bench.sh into the committed evidence:
    python scripts/summarize_profiles.py r01
reads gpurun_out/prof_<tag>/ and writes profiles/<tag>_kernel_stats.csv (the --stats table, top
rows), profiles/<tag>_pmc.json (per-launch FETCH_SIZE / WRITE_SIZE of the dominant kernel, the
calibration factors and the corrected HBM traffic) and profiles/<tag>_summary.md.

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are
KiB, collected in separate --pmc passes; on gfx950 FETCH_SIZE under-reports wide coalesced reads
(128 B requests tallied as 64 B) and WRITE_SIZE is uncalibrated, so both are calibrated on
gn2v::touch_rows_kernel, which moves an exactly known byte count with the same access shape.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BYTES_PER_PAIR = 12288


def one(pattern):
    files = glob.glob(pattern, recursive=True)
    if not files:
        raise SystemExit(f"missing {pattern}")
    return files[0]


def per_kernel_counter(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def pick(agg, needle, exact=None):
    """Counter rows of the kernel named exactly `exact` (the dominant instantiation of the
    --stats table) or, failing that, of the first kernel whose name contains `needle`."""
    if exact is not None:
        for name, vals in agg.items():
            if name == exact or name.split("(")[0] == exact.split("(")[0]:
                return name, vals
    for name, vals in agg.items():
        if needle in name:
            return name, vals
    raise SystemExit(f"kernel {needle} not found")


def main(tag):
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = list(csv.DictReader(open(one(f"{src}/stats/**/*kernel_stats.csv"))))
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in stats[:12]:
            w.writerow([r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                        r["Percentage"], r["MinNs"], r["MaxNs"]])
    needle = "cbow_" if any("cbow_" in r["Name"] for r in stats[:3]) else "sgns_"
    sg = next(r for r in stats if needle in r["Name"])
    wk = next(r for r in stats if "walk_kernel" in r["Name"] or "walk_rec_kernel" in r["Name"])
    avg_ms = float(sg["AverageNs"]) / 1e6
    # like for like with bench.py's HIP events: the timed launches are the last `launches` ones
    # of that kernel in the trace (the warm-up's come first and may be smaller)
    trace = [r for r in csv.DictReader(open(one(f"{src}/stats/**/*kernel_trace.csv")))
             if r["Kernel_Name"] == sg["Name"]]
    trace.sort(key=lambda r: int(r["Start_Timestamp"]))

    fetch = per_kernel_counter(one(f"{src}/fetch/**/*counter_collection.csv"), "FETCH_SIZE")
    write = per_kernel_counter(one(f"{src}/write/**/*counter_collection.csv"), "WRITE_SIZE")
    _, f_vals = pick(fetch, needle, sg["Name"])
    _, w_vals = pick(write, needle, sg["Name"])
    cal_f = per_kernel_counter(one(f"{src}/cal_fetch/**/*counter_collection.csv"), "FETCH_SIZE")
    cal_w = per_kernel_counter(one(f"{src}/cal_write/**/*counter_collection.csv"), "WRITE_SIZE")
    _, cf = pick(cal_f, "touch_rows_kernel")
    _, cw = pick(cal_w, "touch_rows_kernel")
    cal = json.loads([l for l in open(f"{src}/cal_fetch.log") if l.startswith("{")][-1])
    bench = json.loads([l for l in open(f"{src}/stats.log") if l.startswith("{")][-1])

    mean = lambda v: sum(v) / len(v)  # noqa: E731
    timed = trace[-int(bench["roofline"]["launches"]):]
    timed_ms = mean([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in timed])
    fetch_factor = cal["read_bytes_per_launch"] / (mean(cf) * 1024)
    write_factor = cal["write_bytes_per_launch"] / (mean(cw) * 1024)
    # Totals over ALL launches of the run (warm-up and timed steps alike: their launches differ
    # in size when the steps do not fill whole rounds), divided by the launches: per-launch means
    # whose ratio is the ratio of the totals.
    n_calls = int(sg["Calls"])
    raw_f, raw_w = sum(f_vals) * 1024 / n_calls, sum(w_vals) * 1024 / n_calls
    read_b, write_b = raw_f * fetch_factor, raw_w * write_factor
    # algorithmic bytes of the whole run: every step of both phases trains walks x pairs per walk
    # (the bench line holds the timed phase: bytes per launch x launches)
    timed_alg = bench["roofline"]["algorithmic_bytes_per_launch"] * bench["roofline"]["launches"]
    alg = timed_alg * (bench["steps"] + bench["warmup"]) / bench["steps"] / n_calls
    pairs_per_launch = alg / BYTES_PER_PAIR
    out = {
        "tag": tag,
        "command": "bench.py " + " ".join(bench.get("argv", [])),
        "kernel": sg["Name"],
        "launches_profiled": int(sg["Calls"]),
        "avg_launch_ms_rocprof": avg_ms,
        "avg_launch_ms_rocprof_timed_launches": timed_ms,
        "avg_launch_ms_bench_hip_events": bench["roofline"]["avg_launch_ms"],
        "pairs_per_launch": None if needle == "cbow_" else pairs_per_launch,
        "algorithmic_bytes_per_launch": alg,
        "algorithmic_GBps": alg / (avg_ms * 1e-3) / 1e9,
        "frac_of_8TBps": alg / (avg_ms * 1e-3) / 8e12,
        "FETCH_SIZE_KiB_per_launch": raw_f / 1024,
        "WRITE_SIZE_KiB_per_launch": raw_w / 1024,
        "calibration": {
            "kernel": "gn2v::touch_rows_kernel",
            "known_read_bytes": cal["read_bytes_per_launch"],
            "known_write_bytes": cal["write_bytes_per_launch"],
            "FETCH_SIZE_KiB": mean(cf), "WRITE_SIZE_KiB": mean(cw),
            "fetch_factor": fetch_factor, "write_factor": write_factor,
            "touch_rows_GBps": cal["GBps"],
        },
        "hbm_read_bytes_per_launch": read_b,
        "hbm_write_bytes_per_launch": write_b,
        "hbm_traffic_bytes_per_launch": read_b + write_b,
        "hbm_traffic_GBps": (read_b + write_b) / (avg_ms * 1e-3) / 1e9,
        "traffic_over_algorithmic": (read_b + write_b) / alg,
        "walk_kernel_avg_ms": float(wk["AverageNs"]) / 1e6,
        "config": bench["config"],
    }
    with open(os.path.join(dst, f"{tag}_pmc.json"), "w") as f:
        json.dump(out, f, indent=1)
    with open(os.path.join(dst, f"{tag}_summary.md"), "w") as f:
        f.write(f"# rocprofv3 summary `{tag}`\n\n")
        f.write(f"Workload: {bench['config']['workload']} (update mode {bench['config']['update_mode']}).\n\n")
        f.write("| kernel | calls | avg ms | % of GPU time |\n|---|---|---|---|\n")
        for r in stats[:6]:
            f.write(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs']) / 1e6:.3f} | {r['Percentage']} |\n")
        units = (f"{alg / 1e9:.1f} GB algorithmic per launch" if needle == "cbow_"
                 else f"{pairs_per_launch:.0f} pairs per launch")
        f.write(f"\n`{sg['Name'].split('(')[0].replace('void ', '')}`: the {len(timed)} launches of the timed "
                f"region average {timed_ms:.2f} ms in rocprofv3's trace vs {bench['roofline']['avg_launch_ms']:.2f} ms "
                f"by the HIP events of bench.py in the same run.  Over all {n_calls} launches (the warm-up's "
                f"included, smaller when its steps do not fill a round): {units}, {avg_ms:.2f} ms -> "
                f"{out['algorithmic_GBps']:.0f} GB/s algorithmic = {out['frac_of_8TBps']:.3f} of 8 TB/s.\n\n")
        f.write(f"PMC (separate passes): FETCH_SIZE {raw_f / 1024:.0f} KiB, WRITE_SIZE {raw_w / 1024:.0f} KiB per launch (mean over all {n_calls} launches); "
                f"calibration on `touch_rows_kernel` (known bytes): fetch x{fetch_factor:.3f}, write x{write_factor:.3f} -> "
                f"HBM traffic {out['hbm_traffic_bytes_per_launch'] / 1e9:.1f} GB per launch = {out['hbm_traffic_GBps']:.0f} GB/s "
                f"({out['traffic_over_algorithmic']:.3f} x the algorithmic bytes).\n")
    roof = bench.get("roofline", {})
    if roof.get("bound") == "l2_atomic":
        # the resident kernel's own ceiling: f32 atomic adds of the central rows' gradients
        dwords = roof["atomic_dwords_per_pair"] * pairs_per_launch
        out["l2_atomic"] = {
            "atomic_dwords_per_launch": dwords,
            "G_atomic_adds_per_s": dwords / (avg_ms * 1e-3) / 1e9,
            "G_atomic_adds_per_s_timed_launches": dwords * len(timed) / len(timed)
            / (timed_ms * 1e-3) / 1e9 if timed else None,
            "peak_G_atomic_adds_per_s": roof["peak"],
            "frac": dwords / (avg_ms * 1e-3) / 1e9 / roof["peak"],
            "peak_source": "scripts/atomic_probe.hip -> profiles/r05_logs/r5_atomic_probe.log",
        }
        with open(os.path.join(dst, f"{tag}_pmc.json"), "w") as f:
            json.dump(out, f, indent=1)
        with open(os.path.join(dst, f"{tag}_summary.md"), "a") as f:
            f.write(f"\nWhat bounds this kernel is not HBM (0.25 of 8 TB/s above) but the L2 atomic units: every "
                    f"pair adds its gradient to its central row with {roof['atomic_dwords_per_pair']} f32 atomics = "
                    f"{dwords / 1e9:.1f} G adds per launch, {out['l2_atomic']['G_atomic_adds_per_s']:.1f} G adds/s over "
                    f"all launches = {out['l2_atomic']['frac']:.3f} of the {roof['peak']:.0f} G adds/s the chip "
                    f"retires (`scripts/atomic_probe.hip`, `profiles/r05_logs/r5_atomic_probe.log`); bench.py "
                    f"reports this as `roofline.frac` ({roof['frac']:.3f} over the timed launches of this run).\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
