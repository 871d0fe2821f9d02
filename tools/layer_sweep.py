#!/usr/bin/env python3
"""Times representative layers of the nets through demon_bench_layer for every tile / split-K plan.
usage: python tools/layer_sweep.py [--n 32] [--quick]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from demon_amd import DemonContext  # noqa: E402

TILES = ["128x128", "64x128", "32x128", "64x64", "32x64", "32x32", "128x32", "64x32"]
# (label, kind, cin, h, w, cout, kh, kw, sh, sw)
LAYERS = [
    ("conv1y 6->32 9x1 s2", "conv", 6, 192, 256, 32, 9, 1, 2, 1),
    ("conv1x 32->32 1x9 s2", "conv", 32, 96, 256, 32, 1, 9, 1, 2),
    ("conv2_1y 64->64 3x1", "conv", 64, 48, 64, 64, 3, 1, 1, 1),
    ("conv3y 64->128 5x1 s2", "conv", 64, 48, 64, 128, 5, 1, 2, 1),
    ("conv3_1x 128->128 1x3", "conv", 128, 24, 32, 128, 1, 3, 1, 1),
    ("conv4x 256->256 1x5 s2", "conv", 256, 12, 32, 256, 1, 5, 1, 2),
    ("conv4_1y 256->256 3x1", "conv", 256, 12, 16, 256, 3, 1, 1, 1),
    ("conv5_1x 512->512 1x3", "conv", 512, 6, 8, 512, 1, 3, 1, 1),
    ("refine4 up 512->256", "deconv", 512, 6, 8, 256, 4, 4, 2, 2),
    ("refine3 up 514->128", "deconv", 514, 12, 16, 128, 4, 4, 2, 2),
    ("refine2 up 256->64", "deconv", 256, 24, 32, 64, 4, 4, 2, 2),
    ("refine0 up 128->32", "deconv", 128, 96, 128, 32, 4, 4, 2, 2),
    ("predict2 conv1 128->24", "conv", 128, 48, 64, 24, 3, 3, 1, 1),
    ("rf conv1_1 64->64 3x3", "conv", 64, 96, 128, 64, 3, 3, 1, 1),
    ("rf conv2_1 128->128 3x3", "conv", 128, 48, 64, 128, 3, 3, 1, 1),
    ("rf pd0 conv1 64->16", "conv", 64, 192, 256, 16, 3, 3, 1, 1),
    ("motion_fc1 6144->1024", "dense", 6144, 1, 1, 1024, 1, 1, 1, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--quick", action="store_true", help="automatic plan only")
    ap.add_argument("--only", type=str, default="")
    args = ap.parse_args()
    ctx = DemonContext(0, 1)
    for lab, kind, cin, h, w, cout, kh, kw, sh, sw in LAYERS:
        if args.only and args.only not in lab:
            continue
        ms, tf = ctx.bench_layer(kind, args.n, cin, h, w, cout, kh, kw, sh, sw)
        line = "%-26s auto %7.3f ms %6.1f TF/s" % (lab, ms, tf)
        if not args.quick:
            mpad = (cout + 31) // 32 * 32
            for t, name in enumerate(TILES):
                bm = int(name.split("x")[0])
                if mpad % bm:
                    continue
                best = None
                for ks in (1, 2, 4, 8):
                    try:
                        ms, tf = ctx.bench_layer(kind, args.n, cin, h, w, cout, kh, kw, sh, sw, tile=t, ksplit=ks, iters=10)
                    except Exception:
                        continue
                    if best is None or tf > best[1]:
                        best = (ks, tf)
                line += " | %s k%d %5.1f" % (name, best[0], best[1])
        print(line, flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
