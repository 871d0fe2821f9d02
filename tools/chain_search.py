#!/usr/bin/env python3
"""Small batches: does running a stride-1 k x 1 / 1 x k pair as ONE chained launch (plan kinds 6 / 7 on the k x 1 layer) pay inside a whole pass?
The replay tuner compares a chain with its two launches in a replay loop; at batch 1 a pass is ~ 330 dependent launches of 5 - 8 us, so what a chain
saves is a launch boundary -- visible only in the pass.  For every chainable pair and every chained variant the library accepts: install, replay the
whole pass `--replays` times (graph), keep the entry when the pass gets faster by `--margin` twice.
usage: python tools/chain_search.py --batch 1 [--out gpurun_out/plan.json]"""
import argparse, json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import demon_amd.lanes  # noqa
import numpy as np
from demon_amd import DemonContext, weights as W
from demon_amd.engine import DemonError


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--replays", type=int, default=150)
    ap.add_argument("--margin", type=float, default=0.002)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    n = args.batch
    ctx = DemonContext(0, n)
    ctx.set_weights(W.synthetic_weights(seed=1))
    assert ctx.load_tuned_plan(n, nearest=False) == n
    plan_file = ctx.plan_file
    rng = np.random.default_rng(0)
    pair = rng.random((n, 6, 192, 256), dtype=np.float32) - np.float32(0.5)
    ctx.upload_inputs(pair, pair[:, 3:6].reshape(n, 3, 48, 4, 64, 4).mean(axis=(3, 5)).astype(np.float32))

    def rate():
        for _ in range(5):
            ctx.run_full(n, 3)
        ctx.synchronize()
        best = 0.0
        for _ in range(2):
            t0 = time.perf_counter()
            for _ in range(args.replays):
                ctx.run_full(n, 3)
            ctx.synchronize()
            best = max(best, n * args.replays / (time.perf_counter() - t0))
        return best

    cur = ctx.get_plan(n)
    base = rate()
    print("plan %s: %.1f pairs/s" % (plan_file, base), flush=True)
    ylayers = [k for k in cur if k.endswith("_1y")]
    kept = []
    for y in sorted(ylayers):
        best = None
        for kind, nv in ((6, 22), (7, 18)):
            for v in range(nv):
                try:
                    ctx.set_plan(n, {y: [kind, v, 1]})
                except DemonError:
                    continue
                r = rate()
                if best is None or r > best[0]:
                    best = (r, kind, v)
                ctx.set_plan(n, {y: cur[y]})
        if best is None:
            continue
        ref = rate()
        print("  %-24s best chained (%d, %d) %.1f vs %.1f" % (y, best[1], best[2], best[0], ref), flush=True)
        if best[0] > ref * (1 + args.margin):
            ctx.set_plan(n, {y: [best[1], best[2], 1]})
            again = rate()
            ctx.set_plan(n, {y: cur[y]})
            ref2 = rate()
            if again > ref2 * (1 + args.margin):
                cur[y] = [best[1], best[2], 1]
                ctx.set_plan(n, {y: cur[y]})
                kept.append(y)
                print("     kept (%.1f vs %.1f)" % (again, ref2), flush=True)
    final = rate()
    print("final %.1f pairs/s (%d pairs chained: %s); start %.1f" % (final, len(kept), kept, base), flush=True)
    if args.out and kept:
        meta = json.load(open(os.path.join(ROOT, "demon_amd", "tuned", plan_file)))
        meta["plan"] = cur
        json.dump(meta, open(args.out, "w"), indent=0, sort_keys=True)
        print("wrote", args.out)
    ctx.close()


if __name__ == "__main__":
    main()
