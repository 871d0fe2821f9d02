#!/usr/bin/env python3
"""Deep-layer shapes at one batch size: best LDS-tiled plan (im2col / patch) vs the register-streaming kernel variants.
usage: python tools/deep_sweep.py [--n 32]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from demon_amd import DemonContext  # noqa: E402

# (label, kind, cin, h, w, cout, kh, kw, sh, sw)
LAYERS = [
    ("conv2_1y 64->64 3x1 @48x64", "conv", 64, 48, 64, 64, 3, 1, 1, 1),
    ("conv3y 64->128 5x1 s2 @48x64", "conv", 64, 48, 64, 128, 5, 1, 2, 1),
    ("conv3x 128->128 1x5 s2 @24x64", "conv", 128, 24, 64, 128, 1, 5, 1, 2),
    ("conv3_1x 128->128 1x3 @24x32", "conv", 128, 24, 32, 128, 1, 3, 1, 1),
    ("conv4y 128->256 5x1 s2 @24x32", "conv", 128, 24, 32, 256, 5, 1, 2, 1),
    ("conv4x 256->256 1x5 s2 @12x32", "conv", 256, 12, 32, 256, 1, 5, 1, 2),
    ("conv4_1y 256->256 3x1 @12x16", "conv", 256, 12, 16, 256, 3, 1, 1, 1),
    ("conv5y 256->512 5x1 s2 @12x16", "conv", 256, 12, 16, 512, 5, 1, 2, 1),
    ("conv5x 512->512 1x5 s2 @6x16", "conv", 512, 6, 16, 512, 1, 5, 1, 2),
    ("conv5_1x 512->512 1x3 @6x8", "conv", 512, 6, 8, 512, 1, 3, 1, 1),
    ("motion_conv1 512->128 3x3 @6x8", "conv", 512, 6, 8, 128, 3, 3, 1, 1),
    ("refine4 up 512->256 @6x8", "deconv", 512, 6, 8, 256, 4, 4, 2, 2),
    ("refine2 up 256->64 @24x32", "deconv", 256, 24, 32, 64, 4, 4, 2, 2),
    ("pf2 conv1 128->24 3x3 @48x64", "conv", 128, 48, 64, 24, 3, 3, 1, 1),
    ("rf conv2_1 128->128 3x3 @48x64", "conv", 128, 48, 64, 128, 3, 3, 1, 1),
    ("motion_fc1 6144->1024", "dense", 6144, 1, 1, 1024, 1, 1, 1, 1),
]
VARIANTS = ["128x32", "128x64", "64x32", "64x64", "32x32", "32x64", "128x64w", "64x64w", "128x32w", "256x32w",   # w: 64-row wave tiles
            "32x32k4", "32x32k8", "32x32k16", "32x64k8", "64x32wk8", "64x64wk4", "64x64wk8", "64x32k4"]                  # kN: N wave groups split K inside the workgroup


FRAG = ["128x64", "64x64", "128x128", "64x128", "256x32", "64x128b", "128x32", "64x256",   # conv_frag.hip variants
        "64x64s2", "128x32s2", "128x64s2", "64x128bs2", "64x128s2", "32x128s2"]              # s2: two K-steps per barrier


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    ctx = DemonContext.ops_only(0)
    for lab, kind, cin, h, w, cout, kh, kw, sh, sw in LAYERS:
        if args.only and args.only not in lab:
            continue
        mpad = (cout + 31) // 32 * 32
        best_old = None
        for tile in list(range(8)) + [100 + t for t in range(9)]:
            for ks in ((1, 2, 3, 4, 6, 8, 12, 16) if tile < 100 else (0, 2, 4)):
                try:
                    ms, tf = ctx.bench_layer(kind, args.n, cin, h, w, cout, kh, kw, sh, sw, tile=tile, ksplit=ks, iters=10)
                except Exception:
                    continue
                if best_old is None or ms < best_old[0]:
                    best_old = (ms, tf, tile, ks)
        line = "%-32s old best %7.4f ms %6.1f TF (tile %d k%d)" % (lab, best_old[0], best_old[1], best_old[2], best_old[3])
        for v, name in enumerate(VARIANTS):
            if mpad % int(name.split("x")[0]):
                continue
            best = None
            for ks in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48):
                try:
                    ms, tf = ctx.bench_layer(kind, args.n, cin, h, w, cout, kh, kw, sh, sw, tile=200 + v, ksplit=ks, iters=10)
                except Exception:
                    continue
                if best is None or ms < best[0]:
                    best = (ms, tf, ks)
            if best:
                line += " | s%s k%d %6.4f %5.1f" % (name, best[2], best[0], best[1])
        for v, name in enumerate(FRAG):
            if mpad % int(name.split("x")[0]):
                continue
            best = None
            for ks in (1, 2, 3, 4, 6, 8, 12, 16):
                try:
                    ms, tf = ctx.bench_layer(kind, args.n, cin, h, w, cout, kh, kw, sh, sw, tile=300 + v, ksplit=ks, iters=10)
                except Exception:
                    continue
                if best is None or ms < best[0]:
                    best = (ms, tf, ks)
            if best:
                line += " | f%s k%d %6.4f %5.1f" % (name, best[2], best[0], best[1])
        print(line, flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
