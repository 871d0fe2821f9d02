#!/usr/bin/env python3
"""In-pass tuner: ranks the launch-plan candidates of a layer by what the launch costs INSIDE a forward pass, not in a replay loop.

`tools/tune.py` (demon_autotune) times a candidate as back-to-back replays of the one layer: its weights and its input are in the L2 of
every XCD from the replay before.  Inside a pass they are not -- a layer's weights were last read a whole pass ago (183 MB of weights per
pass against 8 x 4 MB of L2), its input was written by another kernel's workgroups into THEIR XCDs' L2 -- and the deep layers (3 .. 8 MB of
weights for 1 536 pixels) are exactly the ones whose ranking changes: round 6's flat line order of conv_wino4 won the replay ranking for
conv5_1 by 0.036 -> 0.027 ms and changed nothing end to end (docs/experiments).  Here every candidate is installed for its layer(s),
`demon_profile_full` runs eager passes with an event pair around every launch, and a candidate's cost is the mean duration of ITS launches in
those passes (split-K reduce launch included).  All target layers take candidate k in the same pass, so a sweep costs
(number of candidates) x (repeats + 1) passes whatever the number of layers.

--lanes L > 1: the same, with L - 1 other contexts replaying whole passes (their shipped throughput-mode plan) beside the profiled context --
the regime of the headline (a launch is timed under contention from the OTHER lanes' different layers, which is what the replay tuner's
throughput mode cannot offer: it runs L copies of the same layer).

usage: python tools/inpass_tune.py [--batch 32] [--lanes 1] [--only conv4,conv5] [--rounds 2] [--out gpurun_out/plan.json]
Writes the plan (same format as tools/tune.py) and prints, per changed layer, the in-pass cost before and after; `--verify S` then times
S graph replays of the old and the new plan, alternating (lanes = 1), or S steps of a lane group (lanes > 1)."""
import argparse
import collections
import json
import os
import re
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import demon_amd.lanes  # noqa: E402,F401   (GPU_MAX_HW_QUEUES before the first HIP call)
import numpy as np  # noqa: E402
from demon_amd import DemonContext, weights as W  # noqa: E402
from demon_amd.engine import DemonError  # noqa: E402

FAMILY = {0: ("conv_mfma<",), 1: ("conv_patch<", "deconv4<"), 4: ("conv_stream<",), 5: ("conv_frag<",), 8: ("wino_deconv<",), 10: ("wino1d<",),
          15: ("wino3rows<",), 16: ("wino4<",)}
SKIP_KINDS = (3, 6, 7, 11, 12, 13, 14)   # small-Cout, chained pairs (and their partners), dense stream, first-layer kernels: left alone


def candidates(ks_list=(1, 2, 3, 4, 6, 8)):
    c = []
    for v in range(14):
        for m in (1, 2, 3):
            c.append((16, v, m))
    for v in range(13):
        for ks in ks_list:
            c.append((10, v, ks))
    for v in range(22):
        for ks in ks_list:
            c.append((5, v, ks))
    for v in range(18):
        for ks in ks_list:
            c.append((4, v, ks))
    for v in range(7):
        for ks in ks_list:
            c.append((8, v, ks))
    for v in range(20):
        c.append((15, v, 1))
    for t in range(9):
        c.append((1, t, 0))
    for t in range(8):
        for ks in ks_list:
            c.append((0, t, ks))
    return c


def tag_matches(cand, tag):
    kind, v, ks = cand
    if not tag.startswith(FAMILY[kind]):
        return False
    if kind in (5, 10, 15, 16) and not re.search(r",v%d[,>]" % v, tag):
        return False
    if kind == 16:
        if (ks == 3) != (",flat>" in tag) or (ks == 2) != (",walk>" in tag):
            return False
    elif kind in (0, 4, 5, 8, 10):
        if (ks > 1) != ("+" in tag):   # a clamped split-K that ran without the split is another candidate's measurement
            return False
    return True


def inputs(n, seed, H=192, W_=256):
    rng = np.random.default_rng(seed)
    pair = rng.random((n, 6, H, W_), dtype=np.float32) - np.float32(0.5)
    return pair, pair[:, 3:6].reshape(n, 3, H // 4, 4, W_ // 4, 4).mean(axis=(3, 5)).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--lanes", type=int, default=1)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--version", type=int, default=1, choices=[1, 2])
    ap.add_argument("--only", default="", help="comma-separated substrings of layer names (default: every layer with a plan entry of a tunable kind)")
    ap.add_argument("--kinds", default="", help="comma-separated plan kinds to draw candidates from (default: all)")
    ap.add_argument("--ks", default="1,2,3,4,6,8", help="split-K values tried on the kernels that have them (small batches: add 12,16,24,32)")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--margin", type=float, default=0.02, help="a candidate replaces the installed entry only if it is faster by this fraction in the confirmation round")
    ap.add_argument("--out", default="")
    ap.add_argument("--verify", type=int, default=0)
    args = ap.parse_args()
    n, L = args.batch, args.lanes
    H, Wd = args.height, args.width
    w = W.synthetic_weights(seed=1, height=H, width=Wd, version=args.version)
    ctx = DemonContext(0, n, H, Wd, version=args.version)
    ctx.set_weights(w)
    assert ctx.load_tuned_plan(n, nearest=False, lanes=L) == n
    plan_file = ctx.plan_file
    ctx.upload_inputs(*inputs(n, 0, H, Wd))
    others = []
    for i in range(1, L):
        o = DemonContext(0, n, H, Wd, version=args.version)
        o.copy_weights_from(ctx) if hasattr(o, "copy_weights_from") else o.set_weights(w)
        o.load_tuned_plan(n, nearest=False, lanes=L)
        o.set_option("side_branches", 0)
        o.upload_inputs(*inputs(n, i, H, Wd))
        others.append(o)
    if L > 1:
        ctx.set_option("side_branches", 0)   # (a lane of a group runs without them, demon_amd/lanes.py)
    base = ctx.get_plan(n)
    pats = [p for p in args.only.split(",") if p]
    targets = [k for k, (kind, _, _) in base.items() if kind not in SKIP_KINDS and (not pats or any(p in k for p in pats))]
    # the 1 x k partner of a chained pair has no launch of its own
    chained = {k[:-1] for k, (kind, _, _) in base.items() if kind in (6, 7)}
    targets = [k for k in targets if k[:-1] not in chained]
    kinds = {int(k) for k in args.kinds.split(",") if k}
    cands = [c for c in candidates(tuple(int(k) for k in args.ks.split(","))) if not kinds or c[0] in kinds]
    print("plan %s, %d target layers, %d candidates, lanes %d" % (plan_file, len(targets), len(cands), L), flush=True)

    def profile(reps):
        for o in others:
            for _ in range(2 + 2 * reps):
                o.run_full(n, 3)
        rec = ctx.profile_full(n, 3, reps)
        for o in others:
            o.synchronize()
        cost = collections.defaultdict(list)
        tags = {}
        for r in rec:
            cost[r["name"]].append(r["ms"])
            tags[r["name"]] = r["kernel"]
        return {k: float(np.mean(v)) for k, v in cost.items()}, tags

    def install(plan):
        ctx.clear_plan(n)
        ctx.set_plan(n, plan)

    cur = dict(base)
    for rnd in range(args.rounds):
        install(cur)
        ref_cost, ref_tags = profile(args.repeats + 2)
        best = {k: (ref_cost.get(k, 1e9), tuple(cur[k]), ref_tags.get(k, "")) for k in targets if k in ref_cost}
        seen = collections.defaultdict(dict)   # layer -> tag -> (ms, cand)
        t0 = time.time()
        for cand in cands:
            trial = dict(cur)
            tried = []
            for k in best:
                try:
                    ctx.set_plan(n, {k: list(cand)})
                    tried.append(k)
                except DemonError:
                    pass
            if not tried:
                continue
            cost, tags = profile(args.repeats)
            for k in tried:
                tag = tags.get(k, "")
                if k in cost and tag_matches(cand, tag):
                    e = seen[k].get(tag)
                    if e is None or cost[k] < e[0]:
                        seen[k][tag] = (cost[k], cand)
            install(cur)
        print("round %d: sweep %.0f s" % (rnd, time.time() - t0), flush=True)
        # confirmation: the three best tags of every layer against the installed entry, more repeats, all layers' k-th choice in one pass
        top = {k: sorted(v.values())[:3] for k, v in seen.items()}
        conf = collections.defaultdict(list)
        for j in range(3):
            trial = dict(cur)
            used = {}
            for k, lst in top.items():
                if j < len(lst):
                    trial[k] = list(lst[j][1])
                    used[k] = lst[j][1]
            install(trial)
            for _ in range(2):
                cost, tags = profile(args.repeats + 2)
                for k, cand in used.items():
                    if k in cost and tag_matches(cand, tags.get(k, "")):
                        conf[k].append((cost[k], cand, tags[k]))
        install(cur)
        again, _ = profile(args.repeats + 2)
        changed = 0
        for k in sorted(best):
            ref = 0.5 * (best[k][0] + again.get(k, best[k][0]))
            by_cand = collections.defaultdict(list)
            for ms, cand, tag in conf.get(k, []):
                by_cand[(cand, tag)].append(ms)
            if not by_cand:
                continue
            (cand, tag), mss = min(by_cand.items(), key=lambda kv: max(kv[1]))
            if max(mss) < ref * (1.0 - args.margin) and list(cand) != list(cur[k]):
                print("  %-34s %-26s %.4f -> %-26s %.4f ms" % (k, best[k][2], ref, tag, max(mss)), flush=True)
                cur[k] = list(cand)
                changed += 1
        print("round %d: %d layers changed" % (rnd, changed), flush=True)
        if not changed:
            break
    if args.out:
        meta = json.load(open(os.path.join(ROOT, "demon_amd", "tuned", plan_file)))
        meta["plan"] = cur
        with open(args.out, "w") as f:
            json.dump(meta, f, indent=0, sort_keys=True)
        print("wrote", args.out)
    if args.verify:
        def rate(plan):
            install(plan)
            for o in others:
                o.clear_plan(n); o.set_plan(n, plan)
            group = [ctx] + others
            for c in group:
                c.run_full(n, 3)
            for c in group:
                c.synchronize()
            t0 = time.perf_counter()
            for i in range(args.verify):
                group[i % len(group)].run_full(n, 3)
            for c in group:
                c.synchronize()
            return n * args.verify / (time.perf_counter() - t0)
        for rep in range(3):
            print("verify: installed plan %.1f pairs/s, in-pass plan %.1f pairs/s" % (rate(base), rate(cur)), flush=True)
    for c in [ctx] + others:
        c.close()


if __name__ == "__main__":
    main()
