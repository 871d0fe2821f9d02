#!/usr/bin/env python3
"""bench.py's host-to-host leg step by step (round 5: it measured 18 ms per step where a fresh context takes 10): demon_full from / to
pageable numpy after each piece of set-up bench.py does before it."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import demon_amd.lanes  # noqa: E402,F401   (GPU_MAX_HW_QUEUES before the first HIP call)
if os.environ.get("E2E_TORCH", "1") == "1":      # bench.py initialises torch's HIP context before the first DemonContext
    import demon_amd  # noqa: F401  (sets GPU_MAX_HW_QUEUES before the runtime starts)
    import torch
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from demon_amd import DemonContext, weights as W  # noqa: E402
from demon_amd.lanes import LaneGroup  # noqa: E402
n = 32
ctx = DemonContext(0, n, 192, 256)
ctx.set_weights(W.synthetic_weights(seed=1))
ctx.load_tuned_plan(n, lanes=5)
rng = np.random.default_rng(0)
pair = rng.random((n, 6, 192, 256), dtype=np.float32) - np.float32(0.5)
img = pair[:, 3:6].reshape(n, 3, 48, 4, 64, 4).mean(axis=(3, 5)).astype(np.float32)
ctx.upload_inputs(pair, img)


def t(tag, reps=5):
    ctx.full(pair, img, 3)
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.full(pair, img, 3)
    print("%-50s %.2f ms" % (tag, 1e3 * (time.perf_counter() - t0) / reps), flush=True)


t("fresh")
g = LaneGroup(first=ctx, lanes=5, batch=n, plan_batch=n)
for c in g.ctxs[1:]:
    c.upload_inputs(pair, img)
g.run_resident(n, len(g), 3); g.synchronize()
t("group created (side branches off)")
g.calibrate(n, 3)
print(g.mapping)
t("calibrated")
g.close()
t("group closed")
ctx.load_tuned_plan(n)
t("latency plan")
ctx.load_tuned_plan(n, lanes=5)
t("throughput plan again")
ctx.profile_full(n, 3, repeats=1)
t("after profile_full")
ctx.set_option("tune_lanes", 4); ctx.profile_full(n, 3, repeats=1); ctx.set_option("tune_lanes", 1)
t("after in-flight profile_full")
def part(tag, fn, reps=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    print("    %-46s %.2f ms" % (tag, 1e3 * (time.perf_counter() - t0) / reps), flush=True)


part("upload_inputs", lambda: ctx.upload_inputs(pair, img))
part("run_full + synchronize", lambda: (ctx.run_full(n, 3), ctx.synchronize()))
part("download_outputs", lambda: ctx.download_outputs(n))
ctx.set_option("side_branches", 0)
part("run_full + synchronize, side branches off", lambda: (ctx.run_full(n, 3), ctx.synchronize()))
ctx.set_option("side_branches", 1)
ctx.release_streams(); ctx.acquire_streams()
t("after a fresh pair of streams")
ctx.close()
