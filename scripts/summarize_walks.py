"""Digest of scripts/profile_walks.sh: python scripts/summarize_walks.py r04_walks
-> profiles/<tag>_kernel_stats.csv and profiles/<tag>_summary.md: per configuration of
scripts/typed_walk_probe.py (a warm-up + 3 timed launches each, in the script's order) the mean
launch duration from rocprofv3's kernel trace and FETCH_SIZE per step from the separate --pmc
pass (raw: the sampler's reads are 4-8 B wide, so sectors are tallied at face value)."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = ["untyped rw.25/ew4", "typed graph unit weights", "first order", "first order + node x2",
           "first order + edge x0.5", "rw.25/ew4 + node x2 + edge x0.5", "rw2/ew.5",
           "rw2/ew.5 typed", "rw4/ew.25", "rw.5/ew2"]
# round 6: the lines above run at the reference's default max_neighbours = 100; then the exact
# walks (None) and the smoke configuration's 10 (scripts/typed_walk_probe.py)
CONFIGS_R6 = CONFIGS + [f"{name}, max_neighbours {mn}" for mn in ("None", "10")
                        for name in ("rw.25/ew4", "first order", "rw2/ew.5",
                                     "rw.25/ew4 + node x2 + edge x0.5")]
STEPS = (1 << 19) * 127


def is_walk(name):
    """walk_kernel (CSR reads) or walk_rec_kernel (edge records)"""
    return "walk_kernel" in name or "walk_rec_kernel" in name


def one(pattern):
    files = glob.glob(pattern, recursive=True)
    if not files:
        raise SystemExit(f"missing {pattern}")
    return files[0]


def main(tag):
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    trace = [r for r in csv.DictReader(open(one(f"{src}/stats/**/*kernel_trace.csv")))
             if is_walk(r["Kernel_Name"])]
    fetch = [r for r in csv.DictReader(open(one(f"{src}/fetch/**/*counter_collection.csv")))
             if is_walk(r["Kernel_Name"]) and r["Counter_Name"] == "FETCH_SIZE"]
    configs = CONFIGS_R6 if len(trace) == 4 * len(CONFIGS_R6) else CONFIGS
    assert len(trace) == 4 * len(configs) == len(fetch), (len(trace), len(fetch))
    stats = list(csv.DictReader(open(one(f"{src}/stats/**/*kernel_stats.csv"))))
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for r in stats[:8]:
            w.writerow([r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                        r["Percentage"]])
    lines = [f"# {tag}: the walk sampler, `gn2v::walk_rec_kernel` / `gn2v::walk_kernel` (BA 10 M / 100 M, 2^19 walks of 128 nodes per launch)",
             "",
             "`rocprofv3 --kernel-trace --stats` and a separate `--pmc FETCH_SIZE` pass of "
             "`python3 scripts/typed_walk_probe.py` (3 timed launches per configuration after a "
             "warm-up; `scripts/profile_walks.sh`, digest by `scripts/summarize_walks.py`).  "
             "FETCH_SIZE raw (KiB x 1024): the sampler's reads are 4-32 B wide.",
             "",
             "| configuration | kernel | ms / launch | steps/s | FETCH B / step | fetch GB/s | 64 B sectors / s |",
             "|---|---|---|---|---|---|---|"]
    for i, name in enumerate(configs):
        rows = trace[4 * i + 1:4 * i + 4]
        ms = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows) / 3 / 1e6
        kib = sum(float(r["Counter_Value"]) for r in fetch[4 * i + 1:4 * i + 4]) / 3
        per_step = kib * 1024 / STEPS
        kernel = rows[0]["Kernel_Name"].split("(")[0].replace("void ", "")
        lines.append(f"| {name} | `{kernel}` | {ms:.2f} | {STEPS / ms * 1e3:.3e} | {per_step:.0f} | "
                     f"{kib * 1024 / ms / 1e6:.0f} | {kib * 1024 / 64 / ms * 1e3:.2e} |")
    open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r04_walks")
