#!/usr/bin/env python3
"""After `bash tools/collect_profiles.sh <tag>` on the GPU box: copies what is to be judged from gpurun_out/<tag>_profiles/ into
profiles/ -- the bench records (the JSON line of each run), the per-launch tables, the rocprofv3 kernel statistics and the PMC
summary, the last two stamped with the hash of the kernel sources + plans they were measured on.

  python tools/finish_profiles.py <tag>
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def json_line(path):
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    if not lines:
        raise SystemExit("no JSON line in %s" % path)
    return lines[-1] + "\n"


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", "%s_profiles" % tag)
    dst = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(src)):
        if name.startswith("bench") and name.endswith(".json"):
            with open(os.path.join(dst, "%s_%s" % (tag, name)), "w") as f:
                f.write(json_line(os.path.join(src, name)))
    for a, b in (("layers_hipevents.txt", "layers_hipevents.txt"), ("layers_config4_hires.txt", "layers_hires_hipevents.txt"), ("layers_v2.txt", "layers_v2_hipevents.txt"), ("layers_batch1.txt", "layers_batch1_hipevents.txt"), ("layers_batch8.txt", "layers_batch8_hipevents.txt"),
                 ("throughput_profile.txt", "throughput_profile.txt")):
        if os.path.exists(os.path.join(src, a)):
            shutil.copyfile(os.path.join(src, a), os.path.join(dst, "%s_%s" % (tag, b)))
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "rocprof_stats.py"), tag, os.path.join(src, "stats")])
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), tag] + [os.path.join(src, d) for d in ("pmc_fetch", "pmc_write", "pmc_mfma")])


if __name__ == "__main__":
    main()
