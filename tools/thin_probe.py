import os, sys
sys.path.insert(0, "/root/repo")
from demon_amd import DemonContext
ctx = DemonContext(0, 32, 192, 256)
for tile, name in ((400, "thin"), (-1, "auto")):
    ms, tf = ctx.bench_layer("conv", 32, 6, 192, 256, 32, 9, 1, 2, 1, tile=tile, ksplit=0, iters=30)
    print(name, ctx.last_kernel(), "%.4f ms %.1f TF/s  %.0f GB/s" % (ms, tf, (32*6*192*256*4 + 32*32*96*256*4)/ms/1e6))
ctx.close()
