#!/usr/bin/env python3
"""Why does the two-context Pipeline (host-to-host!) beat one context on resident inputs?  Same box, same contexts:
  A  K contexts, all steps enqueued up front (multi_ctx_bench "each")
  B  K contexts, at most one step in flight per context (host waits for a context before reusing it)
  C  Pipeline.run_buffers (B + asynchronous H2D / D2H on page-locked buffers)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import demon_amd.lanes  # noqa: E402,F401   (GPU_MAX_HW_QUEUES before the first HIP call)
import numpy as np
from demon_amd import DemonContext, weights as W
from demon_amd.pipeline import Pipeline


def inputs(n, seed):
    rng = np.random.default_rng(seed)
    pair = rng.random((n, 6, 192, 256), dtype=np.float32) - np.float32(0.5)
    return pair, pair[:, 3:6].reshape(n, 3, 48, 4, 64, 4).mean(axis=(3, 5)).astype(np.float32)


n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
w = W.synthetic_weights(seed=1)
if os.environ.get("TORCH") == "1":
    import torch
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
dummies = [DemonContext(0, n) for _ in range(int(os.environ.get("DUMMY", "0")))]   # idle contexts: their streams take hardware queues
print("idle contexts: %d, torch: %s" % (len(dummies), os.environ.get("TORCH", "0")), flush=True)
for k in [int(x) for x in os.environ.get("KS", "1,2,3").split(",")]:
    pipe = Pipeline(w, batch=n, contexts=k)
    ctxs = pipe.ctxs
    for i, c in enumerate(ctxs):
        c.upload_inputs(*inputs(n, i))
    steps = 24
    for mode in "ABAB":
        for i in range(2 * k):
            ctxs[i % k].run_full(n, 3)
        for c in ctxs: c.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            c = ctxs[i % k]
            if mode == "B" and i >= k:
                c.synchronize()
            c.run_full(n, 3)
        for c in ctxs: c.synchronize()
        dt = time.perf_counter() - t0
        print("K=%d mode %s: %.1f pairs/s (%.3f ms per step)" % (k, mode, n * steps / dt, 1e3 * dt / steps), flush=True)
    hb = pipe.buffers(8 * n)
    p, q = inputs(n, 0)
    for i in range(8):
        hb.image_pair[i * n:(i + 1) * n] = p
        hb.image2_2[i * n:(i + 1) * n] = q
    for _ in range(2):
        r = pipe.throughput(hb, 3, repeats=3)
        print("K=%d mode C: %.1f pairs/s (%.3f ms per step)" % (k, r["pairs_per_s"], r["ms_per_batch"]), flush=True)
    hb.release()
    pipe.close()
