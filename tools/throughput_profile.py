#!/usr/bin/env python3
"""Per-launch table of a pass in THROUGHPUT mode: every step timed as L concurrent replays (option tune_lanes; demon_profile_full
then charges a launch 1 / (5 L) of the time 5 launches x L streams take), with the executed-MFMA utilisation next to it.
usage: python tools/throughput_profile.py [--lanes 3] [--batch 32] [--version 1]"""
import argparse, os, re, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import demon_amd.lanes  # noqa: E402,F401   (GPU_MAX_HW_QUEUES before the first HIP call)
import numpy as np
from demon_amd import DemonContext, weights as W

ap = argparse.ArgumentParser()
ap.add_argument("--lanes", type=int, default=4)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--version", type=int, default=1)
args = ap.parse_args()
n = args.batch
ctx = DemonContext(0, n, version=args.version)
ctx.set_weights(W.synthetic_weights(seed=1, version=args.version))
ctx.load_tuned_plan(n, lanes=args.lanes)
rng = np.random.default_rng(0)
pair = rng.random((n, 6, 192, 256), dtype=np.float32) - np.float32(0.5)
ctx.upload_inputs(pair, pair[:, 3:6].reshape(n, 3, 48, 4, 64, 4).mean(axis=(3, 5)).astype(np.float32))
ctx.run_full(n, 3); ctx.synchronize()


def share(tag):
    if tag.startswith("wino_deconv"):
        return 9 / 16
    if tag.startswith("wino3rows<s2"):
        return 9.0 / 12.0
    if tag.startswith("wino3rows<f4") or tag.startswith("wino4<t3"):
        return 0.5
    if tag.startswith("wino4<t5"):
        return 11 / 20
    m = re.match(r"(?:wino1d|wino3rows|conv_row<32x128,)<?t(\d+)", tag)
    if m:
        t = int(m.group(1))
        return 4 / 6 if t == 3 else (t + 2) / (2 * t)
    return 1.0


alone = ctx.profile_full(n, 3, 2)
ctx.set_option("tune_lanes", args.lanes)
conc = ctx.profile_full(n, 3, 2)
tot_a = sum(r["ms"] for r in alone); tot_c = sum(r["ms"] for r in conc)
print("%-44s %-24s %8s %8s %7s %7s" % ("step", "kernel", "alone ms", "thru ms", "util a", "util t"))
agg = {}
for a, c in zip(alone, conc):
    ex = c["flops"] * share(c["kernel"].split("+")[0])
    ua = ex / (a["ms"] * 1e-3) / 157.3e12 if a["ms"] > 0 else 0
    uc = ex / (c["ms"] * 1e-3) / 157.3e12 if c["ms"] > 0 else 0
    print("%-44s %-24s %8.4f %8.4f %7.3f %7.3f" % (c["name"], c["kernel"], a["ms"], c["ms"], ua, uc))
    key = re.sub(r"^net\w+?/", "", c["name"]) if not c["name"].startswith("netRefine") else c["name"]
    e = agg.setdefault(key, [0.0, 0.0, 0.0])
    e[0] += a["ms"]; e[1] += c["ms"]; e[2] += ex
print("\nby layer name over the whole pass (sum of %d launches): alone %.3f ms, throughput mode %.3f ms" % (len(conc), tot_a, tot_c))
for k, (a, c, ex) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-36s alone %7.3f  thru %7.3f  (%4.1f %%)  ideal %7.3f  thru/ideal %5.2f" % (k, a, c, 100 * c / tot_c, ex / 157.3e12 * 1e3, c / max(ex / 157.3e12 * 1e3, 1e-9)))
ctx.close()
