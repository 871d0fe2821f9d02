#!/usr/bin/env python3
"""Split-K with a reduce launch vs combined inside the launch (ksplit + 1000), deep-layer shapes, several batch sizes.
usage: python tools/fixup_sweep.py [--n 1 8 32]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from demon_amd import DemonContext  # noqa: E402

LAYERS = [
    ("conv4_1y 256->256 3x1 @12x16", "conv", 256, 12, 16, 256, 3, 1, 1, 1),
    ("conv5y 256->512 5x1 s2 @12x16", "conv", 256, 12, 16, 512, 5, 1, 2, 1),
    ("conv5_1x 512->512 1x3 @6x8", "conv", 512, 6, 8, 512, 1, 3, 1, 1),
    ("refine4 up 512->256 @6x8", "deconv", 512, 6, 8, 256, 4, 4, 2, 2),
]
PLANS = [("stream 128x32", 200), ("stream 32x32", 204), ("stream 32x32k4", 210), ("frag 128x32", 306), ("frag 64x64", 301), ("mfma 128x32", 6)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs="+", default=[1, 8, 32])
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    ctx = DemonContext.ops_only(0)
    for n in args.n:
        for lab, kind, cin, h, w, cout, kh, kw, sh, sw in LAYERS:
            print("n=%d %s" % (n, lab))
            for name, tile in PLANS:
                line = "   %-16s" % name
                for ks in (1, 2, 4, 8, 16):
                    try:
                        a, _ = ctx.bench_layer(kind, n, cin, h, w, cout, kh, kw, sh, sw, tile=tile, ksplit=ks, iters=args.iters)
                    except Exception:
                        continue
                    if ks == 1:
                        line += "  k1 %6.1f" % (a * 1e3)
                        continue
                    b, _ = ctx.bench_layer(kind, n, cin, h, w, cout, kh, kw, sh, sw, tile=tile, ksplit=ks + 1000, iters=args.iters)
                    line += " | k%d %6.1f / %6.1f us" % (ks, a * 1e3, b * 1e3)
                print(line)


if __name__ == "__main__":
    main()
