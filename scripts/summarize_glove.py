"""Digest of scripts/profile_glove.sh: python scripts/summarize_glove.py r01_glove
-> profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json, profiles/<tag>_summary.md.
FETCH_SIZE / WRITE_SIZE are corrected with the factors calibrated on gn2v::touch_rows_kernel in
the bench profile of the same round (profiles/r01_pmc.json)."""
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_profiles import ROOT, one, per_kernel_counter  # noqa: E402


def main(tag, cal_tag="r01"):
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    stats = list(csv.DictReader(open(one(f"{src}/stats/**/*kernel_stats.csv"))))
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in stats[:14]:
            w.writerow([r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                        r["Percentage"], r["MinNs"], r["MaxNs"]])
    log = open(f"{src}/stats.log").read()
    entries = float(re.search(r"-> ([0-9.e+]+) entries", log).group(1))
    cal = json.load(open(os.path.join(dst, f"{cal_tag}_pmc.json")))
    ff, wf = cal["calibration"]["fetch_factor"], cal["calibration"]["write_factor"]
    fetch = per_kernel_counter(one(f"{src}/fetch/**/*counter_collection.csv"), "FETCH_SIZE")
    write = per_kernel_counter(one(f"{src}/write/**/*counter_collection.csv"), "WRITE_SIZE")
    out = {"tag": tag, "command": "scripts/glove_probe.py --nodes 1000000 --epochs 2",
           "entries_per_launch": entries, "algorithmic_bytes_per_entry": 1088,
           "fetch_calibration_factor": ff, "write_calibration_factor": wf, "kernels": {}}
    lines = []
    for mode, needle in (("write-through", "<2, 0, false>"), ("write-back", "<2, 1, false>"),
                         ("atomic", "<2, 2, false>")):
        st = next((r for r in stats if "glove_kernel" in r["Name"] and needle in r["Name"]), None)
        fv = next((v for k, v in fetch.items() if "glove_kernel" in k and needle in k), None)
        wv = next((v for k, v in write.items() if "glove_kernel" in k and needle in k), None)
        if not (st and fv and wv):
            continue
        ms = float(st["AverageNs"]) / 1e6
        rd = sum(fv) / len(fv) * 1024 * ff
        wr = sum(wv) / len(wv) * 1024 * wf
        alg = entries * 1088
        out["kernels"][mode] = {
            "avg_ms": ms, "calls": int(st["Calls"]), "algorithmic_GBps": alg / ms / 1e6,
            "frac_of_8TBps": alg / ms / 1e6 / 8000, "hbm_read_bytes_per_launch": rd,
            "hbm_write_bytes_per_launch": wr, "traffic_over_algorithmic": (rd + wr) / alg}
        lines.append(f"| {mode} | {ms:.1f} | {alg / ms / 1e6:.0f} | {alg / ms / 1e6 / 8000:.2f} | "
                     f"{rd / entries:.0f} | {wr / entries:.0f} | {(rd + wr) / alg:.2f} |")
    json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
    with open(os.path.join(dst, f"{tag}_summary.md"), "w") as f:
        f.write(f"# {tag}: `gn2v::glove_kernel` (BA 1 M nodes / 10 M edges, walks of 128, window 5, d = 128)\n\n"
                f"`rocprofv3 --kernel-trace --stats` and separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of\n"
                f"`python3 scripts/glove_probe.py --nodes 1000000 --epochs 2`; {entries:.3e} co-occurrence entries\n"
                f"per launch, algorithmic 1 088 B per entry (the 512 B contextual row read and written per entry, the\n"
                f"central row once per record of 16 entries); counters corrected\n"
                f"with the `{cal_tag}` calibration (fetch x{ff:.3f}, write x{wf:.3f}).\n\n"
                "| update mode | avg ms / launch | algorithmic GB/s | of 8 TB/s | HBM read B / entry | HBM written B / entry | traffic / algorithmic |\n"
                "|---|---|---|---|---|---|---|\n" + "\n".join(lines) + "\n")
    print(open(os.path.join(dst, f"{tag}_summary.md")).read())


if __name__ == "__main__":
    main(*(sys.argv[1:] or ["r01_glove"]))
