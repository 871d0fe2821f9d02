#!/usr/bin/env python3
"""BASELINE configs[4] evidence (batch 64 @640x480, "HBM-bound warp2d stress, rocprof GB/s"): HBM bytes per launch from the
rocprofv3 FETCH_SIZE / WRITE_SIZE passes (separate runs; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)
divided by the kernels' average duration from the --kernel-trace --stats run of the same command.

  python tools/hires_counters.py <fetch_dir> <write_dir> <stats_dir> <out.json>
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import kernel_tag  # noqa: E402


def counters(d, name):
    path = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == name:
                agg[kernel_tag(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


def main():
    fdir, wdir, sdir, out = sys.argv[1:5]
    fetch, write = counters(fdir, "FETCH_SIZE"), counters(wdir, "WRITE_SIZE")
    stats = {}
    with open(glob.glob(os.path.join(sdir, "**", "*kernel_stats.csv"), recursive=True)[0]) as f:
        for r in csv.DictReader(f):
            stats[kernel_tag(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]), float(r["Percentage"]))
    res = {}
    for tag in sorted(stats, key=lambda t: -stats[t][2]):
        if tag not in fetch:
            continue
        calls, avg_ns, pct = stats[tag]
        fb = 2048.0 * sum(fetch[tag]) / len(fetch[tag])          # KiB -> bytes, x2 gfx950 correction for wide reads
        wb = 1024.0 * sum(write[tag]) / len(write[tag]) if tag in write else 0.0
        res[tag] = {"calls_in_stats_run": calls, "avg_us": round(avg_ns / 1e3, 2), "time_share_pct": round(pct, 2),
                    "fetch_MB_per_launch_x2": round(fb / 1e6, 3), "write_MB_per_launch": round(wb / 1e6, 3),
                    "hbm_GB_per_s": round((fb + wb) / avg_ns, 1)}
    with open(out, "w") as f:
        json.dump({"workload": "configs[4]: batch 64 @640x480, full pipeline", "note": __doc__.strip().split("\n\n")[0], "kernels": res}, f, indent=1)
    for tag, e in list(res.items())[:14]:
        print("%-36s %8.1f us  %8.2f MB rd  %8.2f MB wr  %7.1f GB/s  %5.1f %%" % (tag, e["avg_us"], e["fetch_MB_per_launch_x2"], e["write_MB_per_launch"], e["hbm_GB_per_s"], e["time_share_pct"]))


if __name__ == "__main__":
    main()
