#!/usr/bin/env python3
"""Times the transposed-conv layers of the nets: best direct plan of the autotuner's families vs the minimal-filtering kernel
(conv_wino.hip) for every variant / split-K.  usage: python tools/wino_probe.py [--n 32]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from demon_amd import DemonContext  # noqa: E402

LAYERS = [("refine4 up 512->256 6x8", 512, 6, 8, 256), ("refine3 up 514->128 12x16", 514, 12, 16, 128), ("refine2 up 258->64 24x32", 258, 24, 32, 64),
          ("refine1 up 128->64 48x64", 128, 48, 64, 64), ("refine0 up 128->32 96x128", 128, 96, 128, 32)]


CONV1D = [("conv2_1y 64->64 3x1 48x64", 64, 48, 64, 64, 3, 1, 1, 1), ("conv2_1x 64->64 1x3 48x64", 64, 48, 64, 64, 1, 3, 1, 1),
          ("conv3_1y 128->128 3x1 24x32", 128, 24, 32, 128, 3, 1, 1, 1), ("conv4_1x 256->256 1x3 12x16", 256, 12, 16, 256, 1, 3, 1, 1),
          ("conv5_1y 512->512 3x1 6x8", 512, 6, 8, 512, 3, 1, 1, 1),
          ("conv3y 64->128 5x1 s2 48x64", 64, 48, 64, 128, 5, 1, 2, 1), ("conv3x 128->128 1x5 s2 24x64", 128, 24, 64, 128, 1, 5, 1, 2),
          ("conv2y 32->32 7x1 s2 96x128", 32, 96, 128, 32, 7, 1, 2, 1), ("conv2x 32->32 1x7 s2 48x128", 32, 48, 128, 32, 1, 7, 1, 2),
          ("conv1x 32->32 1x9 s2 96x256", 32, 96, 256, 32, 1, 9, 1, 2),
          ("predict2 conv1 130->24 3x3", 130, 48, 64, 24, 3, 3, 1, 1), ("predict_depth0 conv1 64->16 3x3", 64, 192, 256, 16, 3, 3, 1, 1),
          ("rf conv1_1 64->64 3x3", 64, 96, 128, 64, 3, 3, 1, 1), ("rf conv2_1 128->128 3x3", 128, 48, 64, 128, 3, 3, 1, 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--skip-deconv", action="store_true")
    ap.add_argument("--only1d", default="", help="substring filter on the 1-D layer labels")
    args = ap.parse_args()
    ctx = DemonContext(0, 1)
    for lab, cin, h, w, cout in ([] if args.skip_deconv else LAYERS):
        line = "%-28s" % lab
        best = None
        for tile in (107, 108):       # the fused four-class kernel (deconv4) tiles
            for ks in (0, 2, 4):
                try:
                    ms, tf = ctx.bench_layer("deconv", args.n, cin, h, w, cout, 4, 4, 2, 2, tile=tile, ksplit=ks, iters=10)
                except Exception:
                    continue
                if best is None or ms < best[0]:
                    best = (ms, tf, tile, ks)
        line += " deconv4 best %.3f ms %5.1f TF/s (t%d k%d)" % best
        for v in range(4):
            wb = None
            allks = []
            for ks in (1, 2, 3, 4, 6, 8):
                try:
                    ms, tf = ctx.bench_layer("deconv", args.n, cin, h, w, cout, 4, 4, 2, 2, tile=400 + v, ksplit=ks, iters=10)
                except Exception:
                    continue
                if not ctx.last_kernel().startswith("wino"):
                    continue
                allks.append("k%d:%.3f" % (ks, ms))
                if wb is None or ms < wb[0]:
                    wb = (ms, tf, ks)
            if wb:
                line += " | wino v%d %.3f ms %5.1f TF/s k%d [%s]" % (v, wb[0], wb[1], wb[2], " ".join(allks))
        print(line, flush=True)
    # separable layers: best direct plan vs 1-D minimal filtering
    for lab, cin, h, w, cout, kh, kw, sh, sw in CONV1D:
        if args.only1d and args.only1d not in lab:
            continue
        line = "%-30s" % lab
        best = None
        cands = [(-1, 0)] + [(100 + t, 0) for t in range(6)] + [(300 + v, ks) for v in (0, 1, 2, 4, 6, 8, 9) for ks in (1, 2)]
        for tile, ks in cands:
            try:
                ms, tf = ctx.bench_layer("conv", args.n, cin, h, w, cout, kh, kw, sh, sw, tile=tile, ksplit=ks, iters=10)
            except Exception:
                continue
            if best is None or ms < best[0]:
                best = (ms, tf, tile, ks)
        line += " direct best %.3f ms %5.1f TF/s (t%d k%d)" % best
        for v in range(8):
            wb = None
            for ks in (1, 2, 4):
                try:
                    ms, tf = ctx.bench_layer("conv", args.n, cin, h, w, cout, kh, kw, sh, sw, tile=400 + v, ksplit=ks, iters=10)
                except Exception:
                    continue
                if not ctx.last_kernel().startswith("wino1d"):
                    continue
                if wb is None or ms < wb[0]:
                    wb = (ms, tf, ks)
            if wb:
                line += " | w1d v%d %.3f ms %5.1f TF/s k%d" % (v, wb[0], wb[1], wb[2])
        print(line, flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
