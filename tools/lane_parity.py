#!/usr/bin/env python3
"""The headline's own protocol held to the CPU oracle: what bench.py does before its timed region -- lane 0 with the weights and the
throughput-mode plan nearest to --lanes, LaneGroup(lanes) on top of it, calibrate() with its defaults (lane counts 1 .. lanes x
placeholder streams, winner verified), the package's hardware-queue request -- then several rounds with every kept lane in flight,
and EVERY pair of EVERY kept lane compared with the oracle (relative L1 <= 1e-3 per key and pair, BASELINE.json's tolerance).
Prints one JSON line.  Test infrastructure (tests/test_fullsize_gpu.py, tests/test_bench_gpu.py run it; the second under
`python -m torch.distributed.run --nproc-per-node 1 ... --dist`, where the process group's and RCCL's streams exist first and
the package asks for 16 hardware queues).
usage: python tools/lane_parity.py [--lanes 5] [--batch 32] [--dist] [--two-procs-tag TAG]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import demon_amd.lanes  # noqa: E402,F401   (GPU_MAX_HW_QUEUES before the first HIP call)

KEYS = ("predict_flow5", "predict_flow2", "predict_depth2", "predict_normal2", "predict_rotation", "predict_translation", "predict_depth0")


def make_inputs(n, seed, height=192, width=256):
    rng = np.random.default_rng(seed)
    pair = rng.random((n, 6, height, width), dtype=np.float32) - np.float32(0.5)
    img2_2 = pair[:, 3:6].reshape(n, 3, height // 4, 4, width // 4, 4).mean(axis=(3, 5)).astype(np.float32)
    return pair, img2_2


def rel_l1(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).sum() / max(np.abs(b).sum(), 1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dist", action="store_true", help="bring up a torch.distributed nccl (= RCCL) process group first, like a rank of the driver's launch")
    ap.add_argument("--no-oracle", action="store_true", help="calibrate, run and report only (the concurrent-calibration test)")
    args = ap.parse_args()
    import torch
    from demon_amd import DemonContext, weights as W
    from demon_amd.lanes import LaneGroup, HW_QUEUES
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if args.dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        t = torch.ones(1, device="cuda")
        dist.all_reduce(t)            # the communicator and its streams really exist
    n = args.batch
    w = W.synthetic_weights(seed=1)
    ctx = DemonContext(local_rank, n, 192, 256)
    ctx.set_weights(w)
    plan_of = ctx.load_tuned_plan(n, lanes=args.lanes)
    group = LaneGroup(first=ctx, lanes=args.lanes, batch=n, device=local_rank, plan_batch=n)
    batches = [make_inputs(n, seed=700 + 13 * i) for i in range(args.lanes)]
    group.upload_inputs(batches)
    group.run_resident(n, len(group), 3)
    group.synchronize()
    rates = group.calibrate(n, 3)
    kept = len(group)
    group.run_resident(n, 3 * kept, 3)          # every kept lane three times, all in flight
    group.synchronize()
    outs = [c.download_outputs(n) for c in group.ctxs]
    group.run_resident(n, 2 * kept, 3)
    group.synchronize()
    stable = all(np.array_equal(c.download_outputs(n)[k], o[k]) for c, o in zip(group.ctxs, outs) for k in KEYS)
    finite = all(np.isfinite(o[k]).all() for o in outs for k in KEYS)
    worst, worst_at, checked = 0.0, None, 0
    if not args.no_oracle:
        from oracle import net_ref
        ref = net_ref.DemonRef(w)
        for lane, (out, (pair, img2_2)) in enumerate(zip(outs, batches)):
            for at in range(0, n, 8):
                want = ref.full(pair[at:at + 8], img2_2[at:at + 8], iterations=3)
                for k in KEYS:
                    for j in range(want[k].shape[0]):
                        e = rel_l1(out[k][at + j], want[k][j])
                        if e > worst:
                            worst, worst_at = e, [lane, at + j, k]
                checked += want[KEYS[0]].shape[0]
    rec = {"lanes_requested": args.lanes, "lanes_kept": kept, "plan_batch_loaded": plan_of, "mapping": group.mapping, "hw_queues": dict(HW_QUEUES),
           "calibration_cells": len(rates), "outputs_finite": bool(finite), "second_round_bit_equal": bool(stable),
           "pairs_checked": checked, "worst_rel_l1": worst, "worst_at": worst_at, "dist": bool(args.dist), "pid": os.getpid()}
    group.close()
    ctx.close()
    if args.dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
