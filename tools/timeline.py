#!/usr/bin/env python3
"""Per-workgroup timeline of single layer launches (diagnostic build of the library).

  python demon_amd/build.py --timeline          # libdemon_hip_tl.so (-DDEMON_TIMELINE)
  DEMON_HIP_LIB=demon_amd/libdemon_hip_tl.so python tools/timeline.py [--batch 32] [--layers a,b,c] [--out file.json]

Every workgroup records wall-clock stamps (100 MHz) at kernel entry, after its prologue (first tiles in LDS), after its K loop
and after its stores have drained, plus where it ran.  The summary answers: how long is a launch from first workgroup start to
last store, how much of that is prologue / K loop / epilogue for a typical workgroup, how many workgroups share a CU, and do
they run in lock-step (all prologues, then all loops, then all store tails at the same time).
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

DEFAULT = ["netFlow2/conv2_1y", "netFlow2/conv3y", "netFlow2/conv3_1x", "netFlow2/conv4_1y", "netFlow2/conv5x", "netFlow2/conv5_1x",
           "netFlow2/refine4/upconv", "netFlow2/refine3/upconv", "netFlow2/refine2/upconv", "netFlow2/predict_flow2/conv1",
           "netRefine/conv1_1", "netRefine/refine0/upconv", "netRefine/predict_depth0/conv1"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--layers", default=",".join(DEFAULT))
    ap.add_argument("--out", default="")
    ap.add_argument("--plan", default="", help="override plans: 'layer=kind,tile,ksplit;layer=...' (kind 4 = streaming kernel)")
    args = ap.parse_args()
    from demon_amd import DemonContext, weights as W
    ctx = DemonContext(0, args.batch, 192, 256)
    ctx.set_weights(W.synthetic_weights(seed=1))
    ctx.load_tuned_plan(args.batch)
    if args.plan:
        ctx.set_plan(args.batch, {kv.split("=")[0]: [int(v) for v in kv.split("=")[1].split(",")] for kv in args.plan.split(";")})
    rng = np.random.default_rng(0)
    pair = rng.random((args.batch, 6, 192, 256), dtype=np.float32) - np.float32(0.5)
    img2_2 = pair[:, 3:6].reshape(args.batch, 3, 48, 4, 64, 4).mean(axis=(3, 5)).astype(np.float32)
    ctx.upload_inputs(pair, img2_2)
    ctx.run_full(args.batch, 3)          # fills every activation buffer with real data
    ctx.synchronize()
    cap = 1 << 16
    rec = np.zeros((cap, 8), np.uint64)
    out = {}
    for layer in args.layers.split(","):
        cnt, ms = ctypes.c_int(), ctypes.c_float()
        kname = ctypes.create_string_buffer(64)
        rc = ctx.lib.demon_debug_timeline(ctx.h, layer.encode(), args.batch, rec.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), cap,
                                          ctypes.byref(cnt), ctypes.byref(ms), kname, 64)
        if rc != 0:
            raise SystemExit("demon_debug_timeline(%s): %s" % (layer, ctx.lib.demon_last_error(ctx.h).decode()))
        r = rec[:cnt.value].astype(np.int64)
        t0 = r[:, 0].min()
        us = lambda x: (x - t0) / 100.0          # 100 MHz ticks -> microseconds since the first workgroup started
        start, pro, loop, end = us(r[:, 0]), us(r[:, 1]), us(r[:, 2]), us(r[:, 3])
        hw, xcc = r[:, 4], r[:, 5] & 0xf
        cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | (xcc << 8)   # cu_id, sh_id, se_id, xcc
        per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
        q = lambda a: [round(float(np.percentile(a, p)), 2) for p in (0, 50, 90, 100)]
        e = {
            "kernel": kname.value.decode(), "hip_event_ms": round(ms.value, 4), "workgroups": int(cnt.value), "cus_used": int(len(per_cu)),
            "wg_per_cu_min_med_max": [int(per_cu.min()), float(np.median(per_cu)), int(per_cu.max())],
            "span_us_first_start_to_last_drain": round(float(end.max()), 2),
            "start_us_p0_50_90_100": q(start), "prologue_us": q(pro - start), "kloop_us": q(loop - pro), "epilogue_us": q(end - loop),
            "wg_lifetime_us": q(end - start),
            # lock-step indicator: fraction of the span during which NO workgroup is inside its K loop
            "no_kloop_active_frac": None,
        }
        grid = np.linspace(0, end.max(), 400)
        active = ((pro[None, :] <= grid[:, None]) & (grid[:, None] < loop[None, :])).sum(axis=1)
        e["no_kloop_active_frac"] = round(float((active == 0).mean()), 3)
        e["kloop_active_wg_profile_10bins"] = [int(active[i * 40:(i + 1) * 40].mean()) for i in range(10)]
        out[layer] = e
        print(layer, json.dumps(e))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)
    ctx.close()


if __name__ == "__main__":
    main()
