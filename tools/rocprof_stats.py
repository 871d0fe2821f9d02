#!/usr/bin/env python3
"""Copies the kernel statistics of a `rocprofv3 --kernel-trace --stats -- python bench.py ...` run into profiles/ and stamps them
with the hash of the kernel sources + shipped plans they were measured on (demon_amd.build.csrc_sha), so that bench.py can put
rocprofv3's average launch duration next to its own hip-event one -- and refuse a summary taken on other kernels.

  python tools/rocprof_stats.py <tag> <rocprof output dir>       ->  profiles/<tag>_bench_kernel_stats.csv (+ .meta.json)
"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
from demon_amd import build as hip_build  # noqa: E402


def main():
    tag, d = sys.argv[1:3]
    found = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))
    if not found:
        raise SystemExit("no *kernel_stats.csv under %s" % d)
    dst = os.path.join(ROOT, "profiles", "%s_bench_kernel_stats.csv" % tag)
    shutil.copyfile(found[-1], dst)
    with open(dst.replace(".csv", ".meta.json"), "w") as f:
        json.dump({"csrc_sha": hip_build.csrc_sha(), "source": os.path.relpath(found[-1], ROOT),
                   "command": "rocprofv3 --kernel-trace --stats -- python bench.py (default workload, graph replay)"}, f, indent=1)
    print(dst)


if __name__ == "__main__":
    main()
