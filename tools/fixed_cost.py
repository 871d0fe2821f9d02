import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from demon_amd import DemonContext
ctx = DemonContext(0, 1)
for plan, name in (("1,3,0", "patch 64x64"), ("1,0,0", "patch 128x128")):
    os.environ["DEMON_FORCE_PLAN"] = plan
    row = []
    for cin in (8, 32, 128, 512):
        ms, tf = ctx.bench_layer("conv", 32, cin, 24, 32, 128, 1, 3, 1, 1, iters=20)
        row.append("Cin%4d %6.1fus" % (cin, ms * 1e3))
    print(name, " | ".join(row), flush=True)
