#!/usr/bin/env python3
"""One layer shape under several forced plans (DEMON_FORCE_PLAN is read at launch time).
usage: python tools/plan_probe.py kind n cin h w cout kh kw sh sw  plan [plan ...]     (plan = "kind,tile,ksplit")"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from demon_amd import DemonContext  # noqa: E402
kind = sys.argv[1]
n, cin, h, w, cout, kh, kw, sh, sw = map(int, sys.argv[2:11])
ctx = DemonContext(0, 1)
for plan in sys.argv[11:]:
    os.environ["DEMON_FORCE_PLAN"] = plan
    ms, tf = ctx.bench_layer(kind, n, cin, h, w, cout, kh, kw, sh, sw, iters=30)
    print("%-12s %.4f ms %6.1f TF/s  [%s]" % (plan, ms, tf, ctx.last_kernel()), flush=True)
ctx.close()
