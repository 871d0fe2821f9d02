#!/usr/bin/env python3
"""What a group of layers costs while several passes are in flight: the lanes rate with the group's steps left out
(DEMON_SKIP_STEPS, wrong results) against the full pass.  One process per group (the hook is read once).
usage: python tools/ablate_lanes.py [--lanes 3] [--batch 32]"""
import argparse, os, subprocess, sys, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
GROUPS = [("none", ""), ("conv1y", "/conv1y"), ("conv1x", "/conv1x"), ("conv2", "/conv2y,/conv2x"), ("conv2_1", "/conv2_1"), ("conv3", "/conv3y,/conv3x"), ("conv3_1", "/conv3_1"),
          ("conv4", "/conv4y,/conv4x"), ("conv4_1", "/conv4_1"), ("conv5", "/conv5y,/conv5x"), ("conv5_1", "/conv5_1"),
          ("refine4", "refine4/upconv"), ("refine3", "refine3/upconv"), ("refine2", "refine2/upconv"),
          ("heads2 conv1", "predict_flow2/conv1,predict_depthnormal2/conv1"), ("heads2 conv2", "predict_flow2/conv2,predict_depthnormal2/conv2"),
          ("extra inputs", "conv2_extra,assemble_inputs"), ("motion", "motion"), ("flow5 head", "predict_flow5,upsample_flow5"),
          ("rf conv0+assemble", "netRefine/conv0,netRefine/assemble"), ("rf conv1", "netRefine/conv1$"), ("rf conv1_1", "netRefine/conv1_1"),
          ("rf conv2", "netRefine/conv2$"), ("rf conv2_1", "netRefine/conv2_1"), ("rf refine1", "netRefine/refine1"), ("rf refine0", "netRefine/refine0"),
          ("rf pd0 conv1", "predict_depth0/conv1"), ("rf pd0 conv2", "predict_depth0/conv2"),
          ("split-K reduces", "@reduce")]     # every conv_splitk_reduce / dense_reduce-free launch of the pass left out (DEMON_SKIP_REDUCE)
ap = argparse.ArgumentParser()
ap.add_argument("--lanes", type=int, default=4)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--child", default=None)
args = ap.parse_args()
if args.child is not None:
    import time
    import numpy as np
    sys.path.insert(0, ROOT)
    from demon_amd import weights as W
    from demon_amd.lanes import LaneGroup
    n = args.batch
    g = LaneGroup(W.synthetic_weights(seed=1), args.lanes, n)
    rng = np.random.default_rng(0)
    for c in g.ctxs:
        pair = rng.random((n, 6, 192, 256), dtype=np.float32) - np.float32(0.5)
        c.upload_inputs(pair, pair[:, 3:6].reshape(n, 3, 48, 4, 64, 4).mean(axis=(3, 5)).astype(np.float32))
    g.run_resident(n, 2 * len(g)); g.synchronize()
    g.calibrate(n, steps_per_lane=2, candidates=[args.lanes])     # the stream -> hardware-queue mapping that is good for this many lanes
    steps = 10 * len(g)
    t0 = time.perf_counter(); g.run_resident(n, steps); g.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"ms_per_step": 1e3 * dt / steps}))
    sys.exit(0)
base = None
for name, subs in GROUPS:
    env = dict(os.environ)
    if subs == "@reduce":
        env["DEMON_SKIP_REDUCE"] = "1"
    elif subs:
        env["DEMON_SKIP_STEPS"] = subs
    r = subprocess.run([sys.executable, __file__, "--child", name, "--lanes", str(args.lanes), "--batch", str(args.batch)], env=env, capture_output=True, text=True)
    try:
        ms = json.loads(r.stdout.strip().splitlines()[-1])["ms_per_step"]
    except Exception:
        print(name, "FAILED", r.stderr[-300:]); continue
    if base is None:
        base = ms
    print("%-20s %7.3f ms per step   group costs %6.3f ms (%4.1f %%)" % (name, ms, base - ms, 100 * (base - ms) / base), flush=True)
