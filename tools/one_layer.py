#!/usr/bin/env python3
"""Runs ONE contraction layer through demon_bench_layer (for rocprofv3 runs of a single kernel variant).
usage: python tools/one_layer.py kind n cin h w cout kh kw sh sw tile ksplit [iters]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from demon_amd import DemonContext  # noqa: E402

kind = sys.argv[1]
n, cin, h, w, cout, kh, kw, sh, sw, tile, ks = map(int, sys.argv[2:13])
iters = int(sys.argv[13]) if len(sys.argv) > 13 else 20
ctx = DemonContext(0, 1)
ms, tf = ctx.bench_layer(kind, n, cin, h, w, cout, kh, kw, sh, sw, tile=tile, ksplit=ks, iters=iters)
print("%s n=%d %d->%d %dx%d tile=%d ks=%d: %.3f ms %.1f TF/s  [%s]" % (kind, n, cin, cout, h, w, tile, ks, ms, tf, ctx.last_kernel()))
ctx.close()
