#!/usr/bin/env python3
"""A/B of the ways to keep several passes in flight on one GPU, all on ONE box in ONE process environment each:
  rr          : the headline's form -- lanes' own graphs fed round robin, stream mapping calibrated (LaneGroup.calibrate)
  rr-nocal    : the same without calibration (streams wherever the runtime put them)
  mask-<L>x<layout>[+<share>] : L lanes, every lane's streams on its own CU mask (demon_set_cu_mask; layout block / stride;
                share 2 = overlapping partitions), round robin, NO calibration
  group-<L>   : ONE hipGraph holding L lanes' passes as parallel branches (demon_lanes_run_group), L steps per launch
Every mode runs in a child process (the stream -> queue mapping depends on every stream the process ever created).
usage: python tools/lane_modes.py [--modes rr,mask-4xblock,...] [--steps 40] [--batch 32] [--repeat 2]"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def child(mode, steps, batch):
    import demon_amd.lanes as L  # noqa: F401   (GPU_MAX_HW_QUEUES before the first HIP call)
    import numpy as np
    from demon_amd import weights as W
    from demon_amd.lanes import LaneGroup, cu_masks
    n = batch
    rng = np.random.default_rng(0)

    def inputs():
        pair = rng.random((n, 6, 192, 256), dtype=np.float32) - np.float32(0.5)
        return pair, pair[:, 3:6].reshape(n, 3, 48, 4, 64, 4).mean(axis=(3, 5)).astype(np.float32)

    kind, _, spec = mode.partition("-")
    if kind == "rr":
        lanes = 4 if spec == "nocal" else 5
    elif kind == "rrside":      # rrside-<L>: L lanes WITH the side branches of a pass on (two streams per lane), stream mapping calibrated for exactly L lanes
        lanes = int(spec)
    elif kind == "pmask":       # pmask-<L>-<P>[c][l]: L lanes over P CU partitions (lane j on partition j % P); c: calibrate the stream mapping; l: latency plan
        lanes = int(spec.split("-")[0])
    else:
        lanes = int((spec.split("x")[0] if kind == "mask" else spec) or 4)
    if kind == "quarter":       # ONE lane on a quarter of the CUs: shows that a mask is honoured by graph replays (expect ~ a third of the rate)
        lanes = 1
    g = LaneGroup(W.synthetic_weights(seed=1), lanes, n)
    if kind == "pmask" and spec.endswith("l"):      # the plan tuned for one pass at a time instead of the throughput-mode one
        g.ctxs[0].load_tuned_plan(n, lanes=1)
        plan = g.ctxs[0].get_plan(n)
        for c in g.ctxs[1:]:
            c.set_plan(n, plan)
    g.upload_inputs([inputs() for _ in range(lanes)])
    note = {}
    if kind == "mask":
        layout, _, share = spec.split("x")[1].partition("+")
        g.set_cu_masks(cu_masks(lanes, layout, int(share or 1)))
    if kind == "quarter":
        g.set_cu_masks([cu_masks(4, "block")[0]])
    if kind == "pmask":
        parts = int(spec.split("-")[1].rstrip("cl"))
        pm = cu_masks(parts, "block")
        g.set_cu_masks([pm[j % parts] for j in range(lanes)])
    g.run_resident(n, len(g), 3)
    g.synchronize()
    if mode == "rr":
        g.calibrate(n, 3)
        note["mapping"] = {k: v for k, v in g.mapping.items() if k != "hw_queues"}
    if kind == "rrside":
        for c in g.ctxs:
            c.set_option("side_branches", 1)
        g.run_resident(n, len(g), 3)
        g.synchronize()
        rates = g.calibrate(n, 3, candidates=[lanes], pads=(0, 1, 2, 3, 4, 5, 6, 7))
        note["mapping"] = {k: v for k, v in g.mapping.items() if k != "hw_queues"}
        note["cells"] = {k: round(v) for k, v in rates.items() if k.startswith("%d@" % lanes)}
        note["side_branches"] = [c.get_option("side_branches") for c in g.ctxs]
    if kind == "pmask" and "c" in spec.split("-")[1]:
        rates = g.calibrate(n, 3, candidates=[lanes], pads=(0, 1, 2, 3, 4, 5, 6, 7))
        note["mapping"] = {k: v for k, v in g.mapping.items() if k != "hw_queues"}
        note["cells"] = {k: round(v) for k, v in rates.items() if k.startswith("%d@" % lanes)}
    k = len(g)

    def run(count):
        if kind == "group":
            g.run_group(n, count // k, 3)
        else:
            g.run_resident(n, count, 3)

    run(2 * k)
    g.synchronize()
    best = 0.0
    for _ in range(3):
        g.synchronize()
        t0 = time.perf_counter()
        run(steps)
        g.synchronize()
        best = max(best, n * (steps // k * k if kind == "group" else steps) / (time.perf_counter() - t0))
    out = g.ctxs[0].download_outputs(n)
    finite = all(np.isfinite(v).all() for v in out.values())
    g.close()
    print(json.dumps(dict(note, mode=mode, lanes=k, pairs_per_s=round(best, 1), finite=bool(finite))), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="rr,rr-nocal,mask-4xblock,mask-4xstride,mask-4xblock+2,mask-2xblock,mask-8xblock+2,group-4,group-3,mask-1xblock,quarter")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--child", default=None)
    args = ap.parse_args()
    if args.child:
        return child(args.child, args.steps, args.batch)
    for _ in range(args.repeat):
        for mode in args.modes.split(","):
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", mode, "--steps", str(args.steps), "--batch", str(args.batch)],
                                   capture_output=True, text=True, timeout=240)
                lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
                print(lines[-1] if lines else json.dumps({"mode": mode, "rc": r.returncode, "stderr": r.stderr[-400:]}), flush=True)
            except subprocess.TimeoutExpired:
                print(json.dumps({"mode": mode, "timeout": True}), flush=True)


if __name__ == "__main__":
    main()
