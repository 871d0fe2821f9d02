#!/usr/bin/env python3
"""demon_full from / to pageable numpy (synchronous copies) on a fresh context: ms per call, and where it goes (upload / run / download).
usage: python tools/e2e_probe.py [batch]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from demon_amd import DemonContext, weights as W  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = DemonContext(0, n, 192, 256)
ctx.set_weights(W.synthetic_weights(seed=1))
ctx.load_tuned_plan(n)
rng = np.random.default_rng(0)
pair = rng.random((n, 6, 192, 256), dtype=np.float32) - np.float32(0.5)
img = pair[:, 3:6].reshape(n, 3, 48, 4, 64, 4).mean(axis=(3, 5)).astype(np.float32)


def t(fn, reps=10):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return 1e3 * (time.perf_counter() - t0) / reps


print("GPU_MAX_HW_QUEUES=%s batch %d" % (os.environ.get("GPU_MAX_HW_QUEUES"), n))
print("  full (host to host)   %.2f ms" % t(lambda: ctx.full(pair, img, 3)))
print("  upload_inputs         %.2f ms" % t(lambda: ctx.upload_inputs(pair, img)))
print("  run_full + sync       %.2f ms" % t(lambda: (ctx.run_full(n, 3), ctx.synchronize())))
print("  download_outputs      %.2f ms" % t(lambda: ctx.download_outputs(n)))
ctx.close()
