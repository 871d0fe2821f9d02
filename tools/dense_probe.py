"""dense_stream.hip on the two big dense layers, every K split and both cache policies, next to the fragment-tiled kernel.
DEMON_DENSE_MODE=1 / 2 (read once per process) runs the diagnostic builds of the loop: loads without MFMAs / MFMAs without loads."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demon_amd import DemonContext

ctx = DemonContext(0, 32, 192, 256)
splits = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 6, 8, 9, 12, 16, 24, 36]
for name, cin, cout in (("dense5", 4608, 4608), ("motion_fc1", 6144, 1024)):
    for v in (0, 1):
        for ks in splits:
            ms, tf = ctx.bench_layer("dense", 32, cin, 1, 1, cout, tile=400 + v, ksplit=ks, iters=30)
            print("%-10s %-28s ks %2d  %.4f ms  %5.1f TF/s  %5.0f GB/s" % (name, ctx.last_kernel(), ks, ms, tf, cin * cout * 4 / ms / 1e6))
    ms, tf = ctx.bench_layer("dense", 32, cin, 1, 1, cout, tile=306, ksplit=12, iters=30)
    print("%-10s %-28s ks 12  %.4f ms" % (name, ctx.last_kernel(), ms))
ctx.close()
