"""conv_row.hip next to the general 1-D kernel on the layers it serves (demon_bench_layer, batch 32)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demon_amd import DemonContext

ctx = DemonContext(0, 32, 192, 256)
for name, cin, h, w, cout, kw in (("conv1x", 32, 96, 256, 32, 9), ("conv2x_it", 32, 48, 128, 32, 7), ("conv1x_v2", 24, 96, 256, 32, 9)):
    for tile in (500, 400, 404):
        ms, tf = ctx.bench_layer("conv", 32, cin, h, w, cout, 1, kw, 1, 2, tile=tile, ksplit=1, iters=30)
        print("%-10s %-22s %.4f ms  %6.1f TF/s  %5.0f GB/s" % (name, ctx.last_kernel(), ms, tf, 32 * 4 * (cin * h * w + cout * h * w // 2) / ms / 1e6))
ctx.close()
