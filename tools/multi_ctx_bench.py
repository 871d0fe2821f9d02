#!/usr/bin/env python3
"""Throughput of K contexts run concurrently on one GPU (separate streams / graphs), device-resident inputs.
  python tools/multi_ctx_bench.py split 32      -> K contexts x (32 / K) pairs   (one batch split over streams)
  python tools/multi_ctx_bench.py each 32       -> K contexts x 32 pairs each    (K steps of the bench in flight)
  optional third argument: comma-separated K list (default 1,2,3,4)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from demon_amd import DemonContext, weights as W


def inputs(n, seed):
    rng = np.random.default_rng(seed)
    pair = rng.random((n, 6, 192, 256), dtype=np.float32) - np.float32(0.5)
    return pair, pair[:, 3:6].reshape(n, 3, 48, 4, 64, 4).mean(axis=(3, 5)).astype(np.float32)


mode = sys.argv[1] if len(sys.argv) > 1 else "each"
total = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ks = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 3, 4]
w = W.synthetic_weights(seed=1)
dummies = [DemonContext(0, 1) for _ in range(int(os.environ.get("DUMMY", "0")))]   # idle contexts: their streams occupy hardware queues
pads = [DemonContext.ops_only(0) for _ in range(int(os.environ.get("PAD1", "0")))]   # one-stream contexts: shift the lanes' streams by one
print("idle contexts: %d, one-stream pads: %d" % (len(dummies), len(pads)), flush=True)
for k in ks:
    n = total // k if mode == "split" else total
    ctxs = []
    for i in range(k):
        c = DemonContext(0, n)
        if ctxs:
            c.copy_weights_from(ctxs[0])
        else:
            c.set_weights(w)
        if not c.load_tuned_plan(n):
            c.autotune(n)
        if "SIDE" in os.environ:
            c.set_option("side_branches", int(os.environ["SIDE"]))
        if "GRAPH" in os.environ:
            c.set_option("hipgraph", int(os.environ["GRAPH"]))
        c.upload_inputs(*inputs(n, i))
        ctxs.append(c)
    for i in range(3 * k):
        ctxs[i % k].run_full(n, 3)
    for c in ctxs: c.synchronize()
    steps = 20 * k
    t0 = time.perf_counter()
    for i in range(steps):
        ctxs[i % k].run_full(n, 3)
    for c in ctxs: c.synchronize()
    dt = time.perf_counter() - t0
    print("contexts %d x batch %d: %.1f pairs/s  (%.3f ms per step of %d pairs)" % (k, n, n * steps / dt, 1e3 * dt / steps, n), flush=True)
    for c in ctxs: c.close()
