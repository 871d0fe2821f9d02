#!/usr/bin/env python3
"""refine4/upconv (512 -> 256 on 6 x 8 maps, blocks_original.py:97-110) at batch n: every minimal-filtering variant x split-K, stand-alone
(demon_bench_layer).  Variant 6 = the reduction split in two inside the workgroup (round 6).  usage: python tools/refine4_probe.py [--n 32]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from demon_amd import DemonContext  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=32)
args = ap.parse_args()
ctx = DemonContext(0, 1)
for lab, cin, h, w, cout in (("refine4 512->256 6x8", 512, 6, 8, 256), ("v2 refine4 480->192 6x8", 480, 6, 8, 192)):
    for v in range(7):
        cells = []
        for ks in (1, 2, 3, 4, 6, 8):
            try:
                ms, tf = ctx.bench_layer("deconv", args.n, cin, h, w, cout, 4, 4, 2, 2, tile=400 + v, ksplit=ks, iters=20)
            except Exception as e:
                cells.append("k%d: %s" % (ks, str(e)[:30]))
                continue
            tag = ctx.last_kernel()
            if not tag.startswith("wino_deconv"):
                continue
            cells.append("k%d %.4f ms (%s)" % (ks, ms, tag))
            if v == 6:
                break
        print("%-26s v%d  %s" % (lab, v, " | ".join(cells)), flush=True)
ctx.close()
