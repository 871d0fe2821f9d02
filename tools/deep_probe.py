"""the deep / small-map layers of one block through every 1-D minimal-filtering shape and split (demon_bench_layer)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demon_amd import DemonContext

ctx = DemonContext(0, 32, 192, 256)
LAYERS = [("conv5_1y", 512, 6, 8, 512, 3, 1, 1, 1), ("conv5_1x", 512, 6, 8, 512, 1, 3, 1, 1), ("conv4_1y", 256, 12, 16, 256, 3, 1, 1, 1),
          ("conv4_1x", 256, 12, 16, 256, 1, 3, 1, 1), ("conv3_1y", 128, 24, 32, 128, 3, 1, 1, 1), ("conv5y", 256, 12, 16, 512, 5, 1, 2, 1),
          ("conv4y", 128, 24, 32, 256, 5, 1, 2, 1)]
for name, cin, h, w, cout, kh, kw, sh, sw in LAYERS:
    res = []
    for v in (int(a) for a in (sys.argv[1:] or "0 1 4 5 8 9 10".split())):
        for ks in (1, 2, 3, 4, 6, 8):
            try:
                ms, tf = ctx.bench_layer("conv", 32, cin, h, w, cout, kh, kw, sh, sw, tile=400 + v, ksplit=ks, iters=20)
            except Exception:
                continue
            tag = ctx.last_kernel()
            if tag.startswith("wino1d<") and (",v%d>" % v) in tag and (ks == 1) == ("+" not in tag):
                res.append((ms, v, ks, tag))
    res.sort()
    print(name, " | ".join("%.4f v%d ks%d" % r[:3] for r in res[:6]))
ctx.close()
