#!/usr/bin/env python3
"""Refines a throughput-mode launch plan ON THE METRIC ITSELF: starting from a base plan, every group of layers (the same layer name in the
four block nets, or one layer of the refinement net) is switched to the entries an alternative plan has for it, the whole headline
configuration is measured (calibrated lane group, resident inputs, `--steps` steps per measurement), and a switch is kept only when it wins
twice against re-measurements of the current plan.  Why: `tools/tune.py --lanes L` ranks a layer's candidates under contention from copies of
the SAME layer; what a pass runs beside is the other lanes' different layers (round 6: a fresh five-round tune lost 3 % to the shipped plan).
usage: python tools/refine_plan.py --base demon_amd/tuned/plan_192x256_n32_l4.json --alt A.json [--alt B.json ...] --out refined.json"""
import argparse
import collections
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import demon_amd.lanes  # noqa: E402,F401   (GPU_MAX_HW_QUEUES before the first HIP call)
import numpy as np  # noqa: E402
from demon_amd import weights as W  # noqa: E402
from demon_amd.lanes import LaneGroup  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base", required=True)
    ap.add_argument("--alt", action="append", default=[])
    ap.add_argument("--out", required=True)
    ap.add_argument("--lanes", type=int, default=4)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--margin", type=float, default=0.003, help="a switch must win by this fraction, twice")
    args = ap.parse_args()
    base = json.load(open(args.base))
    alts = [json.load(open(p))["plan"] for p in args.alt]
    # shape, batch and model follow the base plan's own header
    n = args.batch = int(base.get("batch", args.batch))
    H, Wd, version = int(base.get("height", 192)), int(base.get("width", 256)), int(base.get("model_version", 1))
    g = LaneGroup(W.synthetic_weights(seed=1, height=H, width=Wd, version=version), args.lanes, n, height=H, width=Wd, version=version)
    rng = np.random.default_rng(0)
    for c in g.ctxs:
        pair = rng.random((n, 6, H, Wd), dtype=np.float32) - np.float32(0.5)
        c.upload_inputs(pair, pair[:, 3:6].reshape(n, 3, H // 4, 4, Wd // 4, 4).mean(axis=(3, 5)).astype(np.float32))

    def install(plan):
        for c in g.ctxs:
            c.clear_plan(n)
            c.set_plan(n, plan)

    def measure(plan):
        install(plan)
        g.run_resident(n, 2 * len(g), 3)
        g.synchronize()
        best = 0.0
        for _ in range(2):
            t0 = time.perf_counter()
            g.run_resident(n, args.steps, 3)
            g.synchronize()
            best = max(best, n * args.steps / (time.perf_counter() - t0))
        return best

    cur = dict(base["plan"])
    install(cur)
    g.run_resident(n, len(g), 3)
    g.synchronize()
    g.calibrate(n, 3, candidates=[args.lanes])
    print("mapping", {k: v for k, v in g.mapping.items() if k != "hw_queues"}, flush=True)
    groups = collections.OrderedDict()
    for layer in cur:
        net, _, rest = layer.partition("/")
        key = layer if net == "netRefine" else rest
        groups.setdefault(key, []).append(layer)
    cur_rate = np.median([measure(cur) for _ in range(3)])
    print("base plan %.1f pairs/s, %d groups" % (cur_rate, len(groups)), flush=True)
    kept = []
    for key, layers in groups.items():
        for ai, alt in enumerate(alts):
            cand = dict(cur)
            changed = False
            for layer in layers:
                if layer in alt and list(alt[layer]) != list(cur[layer]):
                    cand[layer] = list(alt[layer])
                    changed = True
            if not changed:
                continue
            r1 = measure(cand)
            if r1 < cur_rate * (1.0 + args.margin):
                print("  %-28s alt %d  %.1f  (current %.1f)" % (key, ai, r1, cur_rate), flush=True)
                continue
            c2 = measure(cur)
            r2 = measure(cand)
            ok = r2 > c2 * (1.0 + 0.66 * args.margin)
            print("  %-28s alt %d  %.1f, again %.1f vs current %.1f -> %s" % (key, ai, r1, r2, c2, "KEPT" if ok else "no"), flush=True)
            if ok:
                cur = cand
                cur_rate = 0.5 * (r1 + r2)
                kept.append((key, ai, [cand[l] for l in layers]))
    final = np.median([measure(cur) for _ in range(3)])
    again = np.median([measure(dict(base["plan"])) for _ in range(3)])
    print("refined plan %.1f pairs/s, base plan re-measured %.1f; %d groups switched: %s" % (final, again, len(kept), [k for k, _, _ in kept]), flush=True)
    out = dict(base, plan=cur, refined_from=os.path.basename(args.base), refined_groups=[k for k, _, _ in kept])
    with open(args.out, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    g.close()


if __name__ == "__main__":
    main()
