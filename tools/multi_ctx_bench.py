#!/usr/bin/env python3
"""Throughput of K contexts x (32/K) pairs run concurrently on one GPU (separate streams / graphs)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from demon_amd import DemonContext, weights as W

def inputs(n, seed):
    rng = np.random.default_rng(seed)
    pair = rng.random((n, 6, 192, 256), dtype=np.float32) - np.float32(0.5)
    return pair, pair[:, 3:6].reshape(n, 3, 48, 4, 64, 4).mean(axis=(3, 5)).astype(np.float32)

w = W.synthetic_weights(seed=1)
total = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for k in (1, 2, 4):
    n = total // k
    ctxs = [DemonContext(0, n) for _ in range(k)]
    for i, c in enumerate(ctxs):
        c.set_weights(w)
        if not c.load_tuned_plan(n):
            c.autotune(n)
        c.upload_inputs(*inputs(n, i))
    for _ in range(3):
        for c in ctxs: c.run_full(n, 3)
    for c in ctxs: c.synchronize()
    steps = 20
    t0 = time.perf_counter()
    for _ in range(steps):
        for c in ctxs: c.run_full(n, 3)
    for c in ctxs: c.synchronize()
    dt = time.perf_counter() - t0
    print("contexts %d x batch %d: %.1f pairs/s  (%.2f ms per %d pairs)" % (k, n, total * steps / dt, 1e3 * dt / steps, total), flush=True)
    for c in ctxs: c.close()
