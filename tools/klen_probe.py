#!/usr/bin/env python3
"""How much of a short-K layer's time is per-launch overhead?  Times a 3x1 conv at levels 2 / 3 with Cin = C and with Cin = 2C (same
output tile count, twice the K loop): T(2C) / T(C) well below 2 means a fused y+x pair (one launch, two K loops) would pay."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from demon_amd import DemonContext  # noqa: E402

ctx = DemonContext.ops_only(0)
n = 32
for lab, c, h, w in (("level 2 (48x64)", 64, 48, 64), ("level 3 (24x32)", 128, 24, 32)):
    for cin in (c, 2 * c, 4 * c):
        best = None
        for tile in [300 + v for v in range(14)] + [200 + v for v in range(10)] + [100 + t for t in range(9)]:
            for ks in (1,):
                try:
                    ms, tf = ctx.bench_layer("conv", n, cin, h, w, c, 3, 1, 1, 1, tile=tile, ksplit=ks, iters=20)
                except Exception:
                    continue
                if best is None or ms < best[0]:
                    best = (ms, tf, tile)
        print("%s  %d -> %d 3x1: best %.1f us  %.1f TF/s (tile %d)" % (lab, cin, c, best[0] * 1e3, best[1], best[2]))
