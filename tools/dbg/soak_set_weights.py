#!/usr/bin/env python3
"""Hunting the silent abort: one process that optionally does what tests/test_distributed.py does on the GPU
(an in-process single-rank RCCL communicator through the C ABI, two spawned ranks sharing the GPU) and then creates a context, loads weights, runs
a pass and closes it again, over and over, for `seconds`.  Prints the elapsed time of every round, so that an abort can be placed.
usage: python -X faulthandler tools/dbg/soak_set_weights.py [--rccl] [--spawn] [--seconds 420] [--batch 8]"""
import argparse, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import demon_amd.lanes  # noqa
import numpy as np
from demon_amd import DemonContext, weights as W
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rccl", action="store_true"); ap.add_argument("--spawn", action="store_true")
    ap.add_argument("--seconds", type=float, default=420.0); ap.add_argument("--batch", type=int, default=8)
    args = ap.parse_args()
    t0 = time.time()
    w = W.synthetic_weights(seed=1)
    if args.rccl:
        from demon_amd import distributed as D
        for _ in range(2):
            ctx = DemonContext(0, 2, 192, 256); comm = D.NativeComm(0, 1, 0)
            ctx.set_weights(w); comm.broadcast_weights(ctx, 0); print("rccl comm count", comm.count(), flush=True)
            comm.close(); ctx.close()
    if args.spawn:
        import tempfile
        import torch.multiprocessing as mp
        import test_distributed as TD
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(TD._gpu_worker, args=(2, TD._free_port(), d), nprocs=2, join=True)
        print("spawned ranks done", flush=True)
    rng = np.random.default_rng(0)
    n = args.batch
    pair = rng.random((n, 6, 192, 256), dtype=np.float32) - np.float32(0.5)
    img = pair[:, 3:6].reshape(n, 3, 48, 4, 64, 4).mean(axis=(3, 5)).astype(np.float32)
    i = 0
    while time.time() - t0 < args.seconds:
        ctx = DemonContext(0, n if i % 3 else 64, 192, 256)
        ctx.set_weights(w)
        ctx.full(pair, img, iterations=3)
        ctx.close()
        i += 1
        print("round %d at %.0f s" % (i, time.time() - t0), flush=True)
    print("finished without an abort after %.0f s, %d rounds" % (time.time() - t0, i))


if __name__ == "__main__":
    main()
