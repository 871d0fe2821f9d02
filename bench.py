#!/usr/bin/env python3
"""bench.py -- image-pairs/s of the full DeMoN forward (bootstrap + 3 x iterative + refine) at 256x192.

  python bench.py --gpus N --steps K --warmup W
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
              --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic pairs already resident in HBM (BASELINE.json
configs[2]: batch 32 per GPU, full pipeline, hipGraph on).  Pairs are independent, so ranks shard the
batch with no data-path collective (weak scaling: 32 pairs per GPU, configs[3] = 8 x 32); the only
collective is one RCCL broadcast of the 183 MB weight blob at start-up, outside the timed region.
Rank 0 prints ONE JSON line.  `roofline` = the dominant kernel family (conv_mfma, fp32 MFMA implicit GEMM)
timed per launch with HIP events on the context stream; `cpu_baseline` = the CPU oracle
("TF-CPU-equivalent" PyTorch-CPU restatement) on this box's host cores, rank 0 / N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_PAIR = 30.353      # 2*MAC of conv/deconv/dense, full pipeline @256x192
# workload -> (height, width, default batch, GFLOP per pair, description)   
WORKLOADS = {
    "full": (192, 256, 32, 30.353, "configs[2]: batch %d/GPU synthetic 256x192 pairs, bootstrap + %d x iterative + refine, device-resident, hipGraph on"),
    "bootstrap": (192, 256, 8, 6.193, "configs[1]: batch %d/GPU synthetic 256x192 pairs, bootstrap net only (netFlow1 + netDM1), device-resident, hipGraph on"),
    "v2": (192, 256, 32, None, "v2 model (python/depthmotionnet/v2, SURVEY 8f row 3): batch %d/GPU synthetic 256x192 pairs, bootstrap + %d x iterative + refine, device-resident, hipGraph on"),
    "hires": (480, 640, 64, 189.70, "configs[4]: batch %d/GPU synthetic 640x480 pairs, bootstrap + %d x iterative + refine, synthetic motion_fc1 38400x1024, hipGraph on"),
}
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


def make_inputs(n, seed, height=192, width=256):
    rng = np.random.default_rng(seed)
    pair = rng.random((n, 6, height, width), dtype=np.float32) - np.float32(0.5)
    img2_2 = pair[:, 3:6].reshape(n, 3, height // 4, 4, width // 4, 4).mean(axis=(3, 5)).astype(np.float32)
    return pair, img2_2


def cpu_baseline(weights, budget_s=20.0, threads=None):
    """Times the CPU oracle (full pipeline, all host cores) on a bounded sample of the same workload."""
    import torch
    from oracle import net_ref
    # threads actually used: torch's default intra-op pool (one per physical core visible to the process),
    # capped at 64 -- oversubscribing the box's 256 logical CPUs made the small convs ~100x slower
    cores = min(torch.get_num_threads(), 64) if threads is None else threads
    torch.set_num_threads(cores)
    ref = net_ref.DemonRef(weights)
    pair, img2_2 = make_inputs(4, seed=100)
    t0 = time.perf_counter()
    ref.full(pair[:1], img2_2[:1], 3)          # warm-up (also pages in oneDNN kernels)
    t_one = time.perf_counter() - t0
    batch = 4 if t_one < budget_s / 8 else 1
    reps = max(1, min(10, int(budget_s / max(t_one * batch, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(reps):
        ref.full(pair[:batch], img2_2[:batch], 3)
    dt = time.perf_counter() - t0
    return {"value": batch * reps / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "%d x batch %d full pipeline (boot + 3 iter + refine) @256x192, PyTorch-CPU fp32 oracle" % (reps, batch)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="pairs per GPU per step (default: the workload's batch)")
    ap.add_argument("--iterations", type=int, default=3)
    ap.add_argument("--workload", choices=["full", "bootstrap", "hires", "v2"], default="full",
                    help="full = BASELINE configs[2] (default, the metric's configuration); bootstrap = configs[1] "
                         "(batch 8, bootstrap net only); hires = configs[4] (batch 64, 640x480, full pipeline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-autotune", action="store_true")
    ap.add_argument("--retune", action="store_true", help="ignore the shipped plan and autotune now")
    ap.add_argument("--reuse-image-features", action="store_true",
                    help="NOT the headline configuration: compute the image-only conv1/conv2 of the iterative nets once per forward "
                         "(loop-invariant hoisting, identical results, 16 launches fewer); reported as a separate metric name")
    ap.add_argument("--layers", action="store_true", help="print the per-launch table to stderr")
    args = ap.parse_args()

    import torch
    from demon_amd import DemonContext, weights as W

    height, width, def_batch, gflop_pair, wl_desc = WORKLOADS[args.workload]
    if args.batch <= 0:
        args.batch = def_batch
    boot_only = args.workload == "bootstrap"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or os.environ.get("DEMON_FORCE_DIST") == "1"   # the env switch exercises the RCCL path with one rank
    if distributed:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))  # RCCL on ROCm
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU fallback for the product path")
    torch.cuda.set_device(local_rank)

    version = 2 if args.workload == "v2" else 1
    ctx = DemonContext(device=local_rank, max_batch=args.batch, height=height, width=width, version=version)
    order = ctx.variables()
    nblob = ctx.blob_size()
    # weights: rank 0 creates the blob, one RCCL broadcast over xGMI puts it on every GPU
    host_weights = None
    t_bcast = 0.0
    if rank == 0:
        host_weights = W.synthetic_weights(seed=1, height=height, width=width, version=version)
        blob = torch.from_numpy(W.weights_to_blob(host_weights, order)).cuda()
    else:
        blob = torch.empty(nblob, dtype=torch.float32, device="cuda")
    if distributed:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dist.broadcast(blob, src=0)
        torch.cuda.synchronize()
        t_bcast = time.perf_counter() - t0
    ctx.set_weights_blob_device(blob.data_ptr(), nblob)
    del blob

    # each rank owns its own shard of the global batch (rank r: pairs [r*B, (r+1)*B))
    pair, img2_2 = make_inputs(args.batch, seed=rank, height=height, width=width)
    n = ctx.upload_inputs(pair, img2_2)
    step = (lambda: ctx.run_bootstrap(n)) if boot_only else (lambda: ctx.run_full(n, args.iterations))
    t0 = time.perf_counter()
    if args.reuse_image_features:
        ctx.set_option("reuse_image_features", 1)
    plan_src = "heuristic"
    if not args.no_autotune:
        # per-layer kernel / tile / split-K selection (untimed set-up): the plan shipped in demon_amd/tuned/ for this
        # workload (measured once on an MI355X by tools/tune.py) or, when there is none, measured now
        if not args.retune and ctx.load_tuned_plan(n):
            plan_src = "demon_amd/tuned"
        else:
            ctx.autotune(n)
            plan_src = "autotune at start-up"
    t_tune = time.perf_counter() - t0

    def barrier():
        if distributed:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = ctx.download_outputs(n, with_depth0=not boot_only)
    finite = all(np.isfinite(v).all() for v in out.values())

    result = None
    if rank == 0:
        pairs = args.batch * world * args.steps
        value = pairs / elapsed
        result = {
            "metric": ("image-pairs/s full 3-iter DeMoN forward @256x192" + (" (image features hoisted out of the iteration loop)" if args.reuse_image_features else "")) if args.workload == "full" else
                      ("image-pairs/s bootstrap net @256x192" if boot_only else
                       ("image-pairs/s full 3-iter DeMoN v2 forward @256x192" if version == 2 else "image-pairs/s full 3-iter DeMoN forward @640x480")),
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl_desc % ((args.batch,) if boot_only else (args.batch, args.iterations)),
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world, "iterations": args.iterations,
                       "sharding": "independent pairs per rank, no data-path collective",
                       "weights": "synthetic He-normal seed 1, RCCL broadcast %.1f ms (untimed)" % (1e3 * t_bcast),
                       "launch_plan": plan_src, "plan_setup_s": round(t_tune, 2)},
            "outputs_finite": bool(finite),
        }
        if gflop_pair:
            result["pipeline_mfma_frac"] = value / world * gflop_pair * 1e9 / (PEAK_FP32_MFMA_TFLOPS * 1e12)
        if not args.no_roofline and not boot_only:
            recs = ctx.profile_full(n, args.iterations, repeats=3)
            conv = [r for r in recs if r["kernel"] == "conv_mfma"]
            ms = sum(r["ms"] for r in conv)
            flops = sum(r["flops"] for r in conv)
            achieved = flops / (ms * 1e-3) / 1e12
            if not gflop_pair:   # no published per-pair figure for this workload: 2*MAC of the launched conv / deconv / dense layers
                result["gflop_per_pair"] = flops / n / 1e9
                result["pipeline_mfma_frac"] = value / world * flops / n / (PEAK_FP32_MFMA_TFLOPS * 1e12)
            result["roofline"] = {
                "kernel": "conv_mfma_kernel (fp32 MFMA implicit GEMM; all conv / deconv / dense launches)",
                "bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                "algorithmic_bytes_per_launch": sum(r["bytes"] for r in conv) / len(conv),
                "launches": len(conv), "avg_launch_ms": ms / len(conv), "flops_per_launch": flops / len(conv),
                "kernel_time_share": ms / sum(r["ms"] for r in recs),
            }
            # HBM bytes per launch cannot be read from inside the process: they come from the rocprofv3 --pmc passes of
            # this same command (FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 x2 read correction) summarised by
            # tools/pmc_summary.py into profiles/<round>_pmc_summary.json
            if args.workload == "full":
                import glob
                pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))
                if pmc:
                    with open(pmc[-1]) as f:
                        c = json.load(f).get("conv", {})
                    if "hbm_traffic_bytes_per_launch" in c:
                        result["roofline"]["traffic"] = c["hbm_traffic_bytes_per_launch"]
                        result["roofline"]["traffic_unit"] = "bytes per launch (PMC, %s)" % os.path.basename(pmc[-1])
                        result["roofline"]["pmc_mfma_busy_frac"] = c.get("mfma_busy_frac")
            if args.layers:
                for r in recs:
                    tf = r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0.0
                    gbs = r["bytes"] / (r["ms"] * 1e-3) / 1e9 if r["ms"] > 0 else 0.0
                    print("%-44s %-16s %8.3f ms %8.2f TF/s %9.1f GB/s" % (r["name"], r["kernel"], r["ms"], tf, gbs), file=sys.stderr)
        if not args.no_cpu_baseline and world == 1 and args.workload == "full":
            result["cpu_baseline"] = cpu_baseline(host_weights)
    ctx.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
