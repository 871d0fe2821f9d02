#!/usr/bin/env python3
"""Autotunes the launch plan (kernel family / tile / split-K per layer) for one workload on the attached MI355X and
writes it to <outdir>/plan_<H>x<W>_n<N>.json (copy into demon_amd/tuned/ to ship it).
usage: python tools/tune.py --height 192 --width 256 --batch 32 [--rounds 3] [--outdir gpurun_out]"""
import argparse, collections, json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import demon_amd.lanes  # noqa: E402,F401   (GPU_MAX_HW_QUEUES before the first HIP call)
from demon_amd import DemonContext, weights as W  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--height", type=int, default=192)
ap.add_argument("--width", type=int, default=256)
ap.add_argument("--batch", type=int, nargs="+", default=[32])
ap.add_argument("--rounds", type=int, default=3, help="majority vote over this many autotune runs")
ap.add_argument("--outdir", default="gpurun_out")
ap.add_argument("--version", type=int, default=1, choices=[1, 2], help="1 = original model, 2 = v2 model")
ap.add_argument("--lanes", type=int, default=1, help="throughput mode: time every candidate as this many concurrent replays (option tune_lanes); the plan is written as ..._l<lanes>.json")
ap.add_argument("--cu-partitions", type=int, default=1, help="tune on 1 / P of the compute units (the context's streams on the CU mask of partition 0, demon_amd.lanes.cu_masks): the regime of a lane group on disjoint CU partitions; one pass at a time; the plan is written as ..._p<P>.json")
ap.add_argument("--only", default="", help="re-tune only the layers whose name contains this substring, on top of the shipped plan of the batch size")
args = ap.parse_args()
os.makedirs(args.outdir, exist_ok=True)
for n in args.batch:
    ctx = DemonContext(0, n, args.height, args.width, version=args.version)
    # real weights and activations in every buffer: all-zero operands clock higher and would bias the comparison
    ctx.set_weights(W.synthetic_weights(seed=1, height=args.height, width=args.width, version=args.version))
    rng = np.random.default_rng(0)
    pair = rng.random((n, 6, args.height, args.width), dtype=np.float32) - np.float32(0.5)
    img2_2 = pair[:, 3:6].reshape(n, 3, args.height // 4, 4, args.width // 4, 4).mean(axis=(3, 5)).astype(np.float32)
    ctx.upload_inputs(pair, img2_2)
    ctx.run_full(n, 3)
    ctx.synchronize()
    if args.cu_partitions > 1:
        from demon_amd.lanes import cu_masks
        ctx.set_cu_mask(cu_masks(args.cu_partitions, "block")[0])
        ctx.run_full(n, 3)
        ctx.synchronize()
    if args.lanes > 1:
        ctx.set_option("tune_lanes", args.lanes)
    if args.only:
        assert ctx.load_tuned_plan(n, nearest=False, lanes=args.lanes, partitions=args.cu_partitions) == n, "no shipped plan to start from"
        os.environ["DEMON_TUNE_ONLY"] = args.only
    votes = collections.defaultdict(collections.Counter)
    for _ in range(args.rounds):
        ctx.autotune(n)
        for layer, p in ctx.get_plan(n).items():
            votes[layer][tuple(p)] += 1
    plan = {layer: list(c.most_common(1)[0][0]) for layer, c in votes.items()}
    path = os.path.join(args.outdir, "plan_%s%dx%d_n%d%s.json" % ("" if args.version == 1 else "v2_", args.height, args.width, n, ("_p%d" % args.cu_partitions) if args.cu_partitions > 1 else ("_l%d" % args.lanes if args.lanes > 1 else "")))
    with open(path, "w") as f:
        json.dump({"gpu": "MI355X (gfx950)", "model_version": args.version, "height": args.height, "width": args.width, "batch": n, "tune_lanes": args.lanes, "cu_partitions": args.cu_partitions, "plan": plan}, f, indent=0, sort_keys=True)
    print(path, len(plan), "layers")
    ctx.close()
