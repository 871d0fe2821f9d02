#!/usr/bin/env python3
"""bench.py -- image-pairs/s of the full DeMoN forward (bootstrap + 3 x iterative + refine) at 256x192.

  python bench.py --gpus N --steps K --warmup W
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
              --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic pairs already resident in HBM (BASELINE.json
configs[2]: batch 32 per GPU, full pipeline, hipGraph on) = ONE hipGraph launch.  Steps go round robin over `--lanes` contexts
of the rank's GPU (demon_amd/lanes.py: own stream, activation arena, resident batch and graph each), so up to that many steps are
in flight and one pass fills the SIMDs another leaves idle at its kernel boundaries; `single_lane` in the same line is the rate of
the same K steps on one lane (one step at a time).  Pairs are independent, so ranks shard the
batch with no data-path collective (weak scaling: 32 pairs per GPU, configs[3] = 8 x 32); the only
collective is one RCCL broadcast of the 183 MB weight blob at start-up, outside the timed region.
Rank 0 prints ONE JSON line.  `roofline` = the single kernel (template instance) with the largest share of the pass by its own time
(rocprofv3's convention: split-K reduce launches listed beside it), `roofline_worst` = the kernel with >= 5 % of the pass that is
furthest below its roofline, `roofline_family` = all conv / deconv / dense launches; all timed per launch with HIP events on the
context stream, with rocprofv3's average of the same kernel next to it when profiles/ holds one for these kernel sources.
In all three, `achieved` / `frac` count the multiply-adds the MATRIX PIPE EXECUTES (<= peak by construction: the minimal-filtering
kernels compute the direct convolution's sums with fewer products), `algorithmic_achieved` / `algorithmic_frac` price the
direct convolution's 2*MAC over the same time; `pipeline_mfma_frac` (algorithmic, whole pass) has
`pipeline_mfma_executed_frac` beside it.
`cpu_baseline` = the CPU oracle ("TF-CPU-equivalent" PyTorch-CPU restatement) on this box's host cores, rank 0 / N=1 only;
`extra` = the PCIe-inclusive host-to-host rates (synchronous pageable copies, and the pinned two-context Pipeline).
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
# lanes on disjoint CU partitions (demon_amd/lanes.py: LaneGroup(partitions=P)) per workload, where that measured faster than lanes that
# share the whole chip; absent = 1
DEFAULT_PARTITIONS = {}


def make_inputs(n, seed, height=192, width=256):
    rng = np.random.default_rng(seed)
    pair = rng.random((n, 6, height, width), dtype=np.float32) - np.float32(0.5)
    img2_2 = pair[:, 3:6].reshape(n, 3, height // 4, 4, width // 4, 4).mean(axis=(3, 5)).astype(np.float32)
    return pair, img2_2


def _cpu_pool_leg():
    """child process of cpu_baseline: the full pipeline on the thread pool (and OpenMP binding) the environment names -- one warm-up
    (never judged: oneDNN creates its primitives per pool size) and two timed runs at batch 1, then the same at batch 8.  Prints a
    JSON line after EVERY stage, so that a pool the parent has to stop (time bound) still reports what it finished."""
    import torch
    from demon_amd import weights as W
    from oracle import net_ref
    t_start = time.perf_counter()
    ref = net_ref.DemonRef(W.synthetic_weights(seed=1))
    pair, img2_2 = make_inputs(8, seed=100)
    torch.set_num_threads(int(os.environ["DEMON_CPU_LEG_THREADS"]))
    rec = {"threads": torch.get_num_threads(), "setup_s": round(time.perf_counter() - t_start, 2)}

    def run(b):
        t0 = time.perf_counter()
        ref.full(pair[:b], img2_2[:b], 3)
        return time.perf_counter() - t0

    rec["batch1_warmup_s"] = round(run(1), 4)
    print(json.dumps(rec), flush=True)
    rec["batch1_s"] = [round(run(1), 4), round(run(1), 4)]
    rec["batch1_pairs_per_s"] = 1.0 / min(rec["batch1_s"])
    print(json.dumps(rec), flush=True)
    run(8)
    rec["batch8_s"] = [round(run(8), 4), round(run(8), 4)]
    rec["batch8_pairs_per_s"] = 8.0 / min(rec["batch8_s"])
    print(json.dumps(rec), flush=True)


def cpu_baseline(weights, budget_s=20.0, pool_timeout_s=25.0):
    """The CPU oracle (PyTorch-CPU restatement of the TF-CPU path, "port") on this box's host cores, after the protocol of
    SURVEY.md section 8(d): batch 1, batch 8 and the metric's own batch 32, warm-ups + timed full-pipeline runs, MEDIAN pairs/s.
    Thread pool: EVERY pool of {32, 64, 128, nproc} is measured WARM in a process of its own (one warm-up run that is never judged --
    oneDNN creates its primitives per pool size -- then two timed runs at batch 1, then the same at batch 8), every pool's numbers are
    in the record (`thread_sweep`), and one more leg runs the best pool with OMP_PROC_BIND=close OMP_PLACES=cores (OpenMP reads its
    binding at start-up: hence the child processes).  The one bound is wall time: a pool gets `pool_timeout_s` seconds; one that has
    not finished by then (on this 256-thread host an oversubscribed pool takes tens of seconds per pair) is stopped and listed with
    the stages it did finish.  The final legs (batch 1 / 8 / 32 medians, bounded to about `budget_s` seconds) run in this process on
    the best pool; value = the best pairs/s any leg measured."""
    import subprocess
    import torch
    from oracle import net_ref
    nproc = os.cpu_count() or 1

    def child(threads, bind):
        env = dict(os.environ, DEMON_CPU_LEG_THREADS=str(threads), OMP_NUM_THREADS=str(threads))
        if bind:
            env.update(OMP_PROC_BIND="close", OMP_PLACES="cores")
        t0 = time.perf_counter()
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-pool-leg"], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        try:
            out, _ = p.communicate(timeout=pool_timeout_s)
            stopped = False
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            stopped = True
        rec = {}
        for line in (out or "").splitlines():
            if line.startswith("{"):
                try:
                    rec = json.loads(line)
                except ValueError:
                    pass
        rec["wall_s"] = round(time.perf_counter() - t0, 1)
        if stopped:
            rec["stopped"] = "not finished within %.0f s: stages missing here were not reached" % pool_timeout_s
        if bind:
            rec["binding"] = "OMP_PROC_BIND=close OMP_PLACES=cores"
        return rec

    sweep = {}
    for c in sorted({min(c, nproc) for c in (32, 64, 128, nproc)}):
        sweep[str(c)] = child(c, False)
    rate = lambda e: max(e.get("batch8_pairs_per_s", 0.0), e.get("batch1_pairs_per_s", 0.0))   # noqa: E731
    cores = int(max(sweep, key=lambda k: rate(sweep[k])))
    bound = child(cores, True)
    ref = net_ref.DemonRef(weights)
    pair, img2_2 = make_inputs(32, seed=100)

    def run(b):
        t0 = time.perf_counter()
        ref.full(pair[:b], img2_2[:b], 3)
        return time.perf_counter() - t0

    torch.set_num_threads(cores)
    run(1)
    spent = time.perf_counter()
    legs = {}
    for b, warm, reps in ((1, 2, 10), (8, 1, 10), (32, 1, 5)):
        for _ in range(warm):
            run(b)
        ts = []
        for _ in range(reps):
            ts.append(run(b))
            if len(ts) >= 3 and time.perf_counter() - spent > budget_s * (1.0 if b < 32 else 1.5):
                break
        legs[b] = {"median_s": float(np.median(ts)), "runs": len(ts), "pairs_per_s": b / float(np.median(ts))}
    best = max(legs, key=lambda b: legs[b]["pairs_per_s"])
    value, what = legs[best]["pairs_per_s"], "batch %d, %d unbound threads" % (best, cores)
    for k, e in sweep.items():
        if rate(e) > value:
            value, what = rate(e), "%s unbound threads (sweep leg)" % k
    if rate(bound) > value:
        value, what = rate(bound), "%d threads bound to cores" % cores
    return {"value": value, "unit": "pairs/s", "cores": cores, "nproc": nproc, "kind": "port",
            "batch1_pairs_per_s": legs[1]["pairs_per_s"], "batch8_pairs_per_s": legs[8]["pairs_per_s"], "batch32_pairs_per_s": legs[32]["pairs_per_s"],
            "thread_sweep": sweep,        # every pool: warm batch-1 and batch-8 times (two timed runs each after a warm-up), or how far it got in its time bound
            "bound_leg": bound,
            "sample": "full pipeline (boot + 3 iter + refine) @256x192, PyTorch-CPU fp32 oracle; pools 32 / 64 / 128 / %d threads each measured warm at "
                      "batch 1 and batch 8 in a process of its own (%.0f s bound per pool), the best pool again with OMP_PROC_BIND=close, then medians of "
                      "%d / %d / %d runs at batch 1 / 8 / 32 on %d threads of %d logical CPUs; value = %s"
                      % (nproc, pool_timeout_s, legs[1]["runs"], legs[8]["runs"], legs[32]["runs"], cores, nproc, what)}


from demon_amd.kernel_names import rocprof_kernel_name  # noqa: E402
from demon_amd import lanes as _lanes  # noqa: E402,F401   (exports GPU_MAX_HW_QUEUES before torch / HIP initialise: demon_amd/lanes.py)


def rocprof_stats():
    """{profile tag: {"avg_ms", "calls", "source"}} from the newest profiles/*_bench_kernel_stats.csv (rocprofv3 --kernel-trace
    --stats of this command), only when its side file says it was taken on the kernel sources and plans of this tree"""
    import csv
    import glob
    from demon_amd import build as hip_build
    from demon_amd.kernel_names import kernel_tag
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench_kernel_stats.csv")))
    if not paths:
        return {}
    meta_path = paths[-1].replace(".csv", ".meta.json")
    try:
        with open(meta_path) as f:
            if json.load(f).get("csrc_sha") != hip_build.csrc_sha():
                return {}
    except (OSError, ValueError):
        return {}
    out = {}
    with open(paths[-1]) as f:
        for row in csv.DictReader(f):
            e = out.setdefault(kernel_tag(row["Name"]), {"ns": 0.0, "calls": 0})
            e["ns"] += float(row["TotalDurationNs"]); e["calls"] += int(row["Calls"])
    return {t: {"avg_ms": e["ns"] / e["calls"] * 1e-6, "calls": e["calls"], "source": os.path.basename(paths[-1])} for t, e in out.items() if e["calls"]}


def flush_c_stdio():
    """fflush(NULL): text that native libraries (RCCL's banner) printed through C stdio must not surface after the JSON line"""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


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
    ap.add_argument("--no-hoisted-leg", action="store_true", help="skip the extra timed leg with reuse_image_features on (value_image_features_hoisted)")
    ap.add_argument("--layers", action="store_true", help="print the per-launch table to stderr")
    ap.add_argument("--weights-bcast", choices=["rccl", "torch"], default="rccl",
                    help="N > 1: rccl = one ncclBroadcast of the packed weight slab through the C ABI (default); torch = "
                         "torch.distributed.broadcast of the TF-layout blob")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-to-host (H2D + kernels + D2H) measurement")
    ap.add_argument("--plan-lanes", type=int, default=0,
                    help="which shipped launch plan to load: 1 = the plan tuned for one pass at a time, > 1 = the throughput-mode plan "
                         "(tools/tune.py --lanes).  Default: what --lanes implies.  `--lanes 1 --plan-lanes 3` runs the headline's kernels one "
                         "pass at a time (what the rocprofv3 / PMC passes of tools/collect_profiles.sh use: counters serialise kernels anyway)")
    ap.add_argument("--lanes", type=int, default=0,
                    help="contexts per GPU the steps go round (each with its own stream, arena, resident batch and hipGraph): that many "
                         "steps are in flight at a time.  Default 0 = measure 1 .. 5 lanes at start-up (untimed set-up) and keep the best; "
                         "1 = one step at a time")
    ap.add_argument("--max-lanes", type=int, default=5, help="--lanes 0: the largest lane count the start-up calibration tries")
    ap.add_argument("--partitions", type=int, default=-1,
                    help="P > 1: the lanes run on P disjoint compute-unit partitions (every lane's streams on the CU mask of partition lane %% P, "
                         "demon_set_cu_mask; --lanes 0 then means one lane per partition); 1: every lane on the whole chip; -1: the shipped default")
    args = ap.parse_args()

    import torch
    from demon_amd import DemonContext, weights as W

    height, width, def_batch, gflop_pair, wl_desc = WORKLOADS[args.workload]
    if args.batch <= 0:
        args.batch = def_batch
    if args.partitions < 0:
        args.partitions = DEFAULT_PARTITIONS.get(args.workload, 1) if args.lanes <= 0 else 1
    auto_lanes = args.lanes <= 0
    if auto_lanes:
        args.lanes = args.partitions if args.partitions > 1 else (3 if args.workload == "hires" else args.max_lanes)
    if args.partitions > 1 and args.lanes % args.partitions:
        raise SystemExit("--lanes must be a multiple of --partitions")
    boot_only = args.workload == "bootstrap"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or os.environ.get("DEMON_FORCE_DIST") == "1"   # the env switch exercises the RCCL path with one rank
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU fallback for the product path")
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))  # RCCL on ROCm

    # host placement (SURVEY 8e "watch"): under a launcher every rank binds its host threads -- and thereby the first-touch pages of the
    # pinned Pipeline buffers allocated later -- to the NUMA node of ITS GPU; a single plain process stays where the OS put it unless
    # DEMON_BIND_NUMA=1 (DEMON_BIND_NUMA=0 switches it off everywhere).  The original mask comes back before the CPU baseline.
    from demon_amd import distributed as D
    affinity0 = os.sched_getaffinity(0)
    want_bind = os.environ.get("DEMON_BIND_NUMA", "1" if ("LOCAL_RANK" in os.environ and world > 1) else "0") == "1"
    placement = D.bind_to_gpu_numa_node(local_rank, apply=want_bind)

    version = 2 if args.workload == "v2" else 1
    ctx = DemonContext(device=local_rank, max_batch=args.batch, height=height, width=width, version=version)
    # weights: rank 0 creates them; ONE RCCL broadcast over xGMI puts them on every GPU.  Default route: the
    # product's own C-ABI path (demon_comm_* + demon_broadcast_weights: packed device slab, no host staging on receivers);
    # --weights-bcast torch = torch.distributed.broadcast of the TF-layout blob + demon_set_weights_blob_device
    host_weights = W.synthetic_weights(seed=1, height=height, width=width, version=version) if rank == 0 else None
    t_bcast, bcast_desc, bcast_route, rccl_nranks = 0.0, "single process: no broadcast", "none", None
    if distributed:
        t_bcast, bcast_desc = D.distribute_weights(ctx, host_weights, rank, world,
                                                   route="rccl" if args.weights_bcast == "rccl" else "torch-gpu")
        # what RCCL itself says (ncclCommCount through the C ABI): a SCALE record with rccl_nranks == n_gpus proves the slab
        # really crossed a communicator of that many ranks
        bcast_route, rccl_nranks = D.LAST_BROADCAST.get("route"), D.LAST_BROADCAST.get("rccl_nranks")
        flush_c_stdio()
    else:
        ctx.set_weights(host_weights)

    # each rank owns its own shard of the global batch (rank r: pairs [r*B, (r+1)*B))
    pair, img2_2 = make_inputs(args.batch, seed=rank, height=height, width=width)
    n = ctx.upload_inputs(pair, img2_2)
    t0 = time.perf_counter()
    if args.reuse_image_features:
        ctx.set_option("reuse_image_features", 1)
    plan_src = "heuristic"
    plan_lanes = args.plan_lanes or args.lanes
    if not args.no_autotune:
        # per-layer kernel / tile / split-K selection (untimed set-up): the plan shipped in demon_amd/tuned/ for this
        # workload (measured once on an MI355X by tools/tune.py) or, when there is none, measured now
        src_n = 0 if args.retune else ctx.load_tuned_plan(n, lanes=plan_lanes, partitions=args.partitions)
        if src_n:
            plan_src = "demon_amd/tuned" if src_n == n else "demon_amd/tuned (plan of batch %d, nearest tuned size)" % src_n
        else:
            ctx.autotune(n)
            plan_src = "autotune at start-up"
    # the lanes: lane 0 is this rank's context; the others get the packed weights device to device and a batch of their own
    # (untimed set-up, like the plan: it also captures every lane's forward graph)
    from demon_amd.lanes import LaneGroup
    group = LaneGroup(first=ctx, lanes=args.lanes, batch=args.batch, height=height, width=width, device=local_rank, version=version, plan_batch=n,
                      partitions=args.partitions)
    for li, c in enumerate(group.ctxs[1:], 1):
        if args.reuse_image_features:
            c.set_option("reuse_image_features", 1)
        c.upload_inputs(*make_inputs(args.batch, seed=1000 * li + rank, height=height, width=width))
    group.run_resident(n, len(group), args.iterations, boot_only)
    group.synchronize()
    lane_rates, lane_mapping = None, None
    if auto_lanes:   # which lane count pays off depends on the runtime's stream -> hardware-queue mapping in THIS process: measured
        lane_rates = group.calibrate(n, args.iterations, boot_only)
        args.lanes, lane_mapping = len(group), group.mapping
    t_tune = time.perf_counter() - t0

    def barrier():
        if distributed:
            dist.barrier()

    def timed(run_steps, sync):
        """W untimed + exactly K timed steps, bracketed by a barrier + device synchronisation on both sides; MAX over the ranks"""
        run_steps(args.warmup)
        sync()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(args.steps)
        sync()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        timed.local_s = dt                  # this rank's own clock (per_rank below); the reported time is the MAX over the ranks
        if distributed:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    elapsed = timed(lambda k: group.run_resident(n, k, args.iterations, boot_only), group.synchronize)
    local_elapsed = timed.local_s
    out = ctx.download_outputs(n, with_depth0=not boot_only)
    finite = all(np.isfinite(v).all() for v in out.values())
    for c in group.ctxs[1:]:
        finite = finite and all(np.isfinite(v).all() for v in c.download_outputs(n, with_depth0=not boot_only).values())
    # The same K steps with the image-only layers of the iterative nets hoisted out of the iteration loop (option reuse_image_features:
    # conv1 / conv2 of netFlow2 / netDM2 see image_pair and their weights only, so one evaluation serves all iterations of a pass --
    # bit-identical outputs, 16 launches and 1.77 GFLOP per pair fewer).  NOT the headline: `value` executes every layer of every
    # iteration, like the reference's three IterativeNet.eval calls (examples/example.py:87-99) do; this is what a caller of demon_full gets
    # by setting the option.
    elapsed_hoisted, hoisted_identical = None, None
    if not boot_only and args.iterations > 1 and not args.reuse_image_features and not args.no_hoisted_leg:
        for c in group.ctxs:
            c.set_option("reuse_image_features", 1)
        group.run_resident(n, len(group), args.iterations, boot_only)   # (every lane captures the other graph outside the timed region)
        group.synchronize()
        elapsed_hoisted = timed(lambda k: group.run_resident(n, k, args.iterations, boot_only), group.synchronize)
        hout = ctx.download_outputs(n, with_depth0=True)
        hoisted_identical = all(np.array_equal(out[k], hout[k]) for k in out)
        for c in group.ctxs:
            c.set_option("reuse_image_features", 0)
    group.close()          # lane 0 = ctx stays (and gets its side branches back)
    shipped_plans = not args.no_autotune and not args.retune
    if args.lanes > 1 and shipped_plans:
        ctx.load_tuned_plan(n)   # ... and the plan tuned for one pass at a time, when the lanes ran a throughput-mode plan
    # the same K steps one at a time on one lane (round 1-3's protocol), for comparison
    # (re-measured even when the calibration kept ONE lane, unless that lane already ran the latency plan with its side branches on)
    single_is_headline = args.lanes == 1 and not (shipped_plans and plan_lanes > 1) and ctx.get_option("side_branches") == 1
    if distributed:   # timed() holds collectives: every rank re-times or none does (a rank whose calibration kept one lane must not skip alone)
        flag = torch.tensor([1 if single_is_headline else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        single_is_headline = bool(flag.item())
    elapsed_single = elapsed if single_is_headline else timed(
        lambda k: [ctx.run_bootstrap(n) if boot_only else ctx.run_full(n, args.iterations) for _ in range(k)], ctx.synchronize)
    local_single = timed.local_s
    # every rank's own numbers (its clock, its calibration): the first multi-GPU run must be diagnosable from the one JSON line
    mine = {"rank": rank, "pairs_per_s": args.batch * args.steps / local_elapsed, "single_lane_pairs_per_s": args.batch * args.steps / local_single,
            "lanes": args.lanes, "lanes_mapping": lane_mapping, "host_placement": placement,
            "lanes_calibration_pairs_per_s": {str(k): round(v, 1) for k, v in lane_rates.items()} if lane_rates else None}
    per_rank = [mine]
    if distributed:
        try:
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)
            per_rank = gathered
        except Exception as e:      # diagnostics must never cost the run its JSON line
            mine["gather_error"] = repr(e)[:200]
    out1 = ctx.download_outputs(n, with_depth0=not boot_only)
    if args.lanes > 1 and shipped_plans:
        ctx.load_tuned_plan(n, lanes=plan_lanes, partitions=args.partitions)   # the per-launch profile below is of the kernels the headline ran
    # (bit-identical when both runs used the same launch plan; a throughput-mode plan sums in another order)
    lanes_vs_single = max(float(np.abs(out[k].astype(np.float64) - out1[k]).sum() / max(np.abs(out1[k]).sum(), 1e-30)) for k in out)

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
                       "lanes": args.lanes, "steps_in_flight_per_gpu": args.lanes, "cu_partitions": args.partitions, "launch_plan_file": getattr(ctx, "plan_file", None),
                       "lanes_calibration_pairs_per_s": {str(k): round(v, 1) for k, v in lane_rates.items()} if lane_rates else None,   # "lanes@placeholder streams"
                       "lanes_mapping": lane_mapping,
                       "hw_queues": dict(_lanes.HW_QUEUES),   # GPU_MAX_HW_QUEUES as this process saw it, who set it, and whether that was in time
                       "sharding": "independent pairs per rank, no data-path collective",
                       "weights": "synthetic He-normal seed 1; %s, %.1f ms (untimed)" % (bcast_desc, 1e3 * t_bcast),
                       "weights_broadcast_ms": round(1e3 * t_bcast, 2), "weights_broadcast_route": bcast_route,
                       "rccl_nranks": rccl_nranks,
                       "launch_plan": plan_src, "plan_setup_s": round(t_tune, 2)},
            "outputs_finite": bool(finite),
            # one batch at a time on one lane (rounds 1-3's protocol, BASELINE configs[2] read strictly): the like-for-like trend number
            "value_single_lane": pairs / elapsed_single,
            "per_rank": {"pairs_per_s_min": min(r["pairs_per_s"] for r in per_rank), "pairs_per_s_max": max(r["pairs_per_s"] for r in per_rank),
                         "single_lane_pairs_per_s_min": min(r["single_lane_pairs_per_s"] for r in per_rank),
                         "lanes": [r["lanes"] for r in per_rank], "ranks": per_rank},
            "single_lane": {"pairs_per_s": pairs / elapsed_single, "ms_per_step": 1e3 * elapsed_single / args.steps,
                            "lane0_outputs_rel_l1": lanes_vs_single, "lane0_outputs_match": bool(lanes_vs_single <= 1e-4),
                            "note": "the same K steps one at a time on one lane (side branches of the pass on a second stream, as in rounds 1-3)"},
        }
        if gflop_pair:
            result["pipeline_mfma_frac"] = value / world * gflop_pair * 1e9 / (PEAK_FP32_MFMA_TFLOPS * 1e12)
        if elapsed_hoisted:
            result["value_image_features_hoisted"] = pairs / elapsed_hoisted
            result["image_features_hoisted"] = {
                "pairs_per_s": pairs / elapsed_hoisted, "ms_per_step": 1e3 * elapsed_hoisted / args.steps, "outputs_bit_identical": bool(hoisted_identical),
                "note": "NOT the headline: the same lanes and steps with option reuse_image_features (conv1 / conv2 of the iterative nets depend on image_pair "
                        "and weights only and are evaluated once per pass instead of once per iteration: 16 launches fewer); `value` executes every layer "
                        "of every iteration like the reference's three eval calls"}
        if not args.no_roofline and not boot_only:
            recs = ctx.profile_full(n, args.iterations, repeats=3)
            # the same launches "in flight": every step timed as L concurrent replays on L streams (demon_profile_full under option
            # tune_lanes; a launch is charged 1 / (5 L) of the time 5 launches x L streams take) -- what a launch costs in the regime
            # the headline runs in.  The stand-alone times above are what the rocprofv3 summary of profiles/ must agree with.
            in_flight = {}
            if args.lanes > 1:
                ctx.set_option("tune_lanes", args.lanes)
                for r in ctx.profile_full(n, args.iterations, repeats=2):
                    e = in_flight.setdefault(r["kernel"].split("+")[0], {"ms": 0.0, "flops": 0.0, "launches": 0})
                    e["ms"] += r["ms"]; e["flops"] += r["flops"]; e["launches"] += 1
                ctx.set_option("tune_lanes", 1)
            conv = [r for r in recs if r["flops"] > 0]          # every conv / deconv / dense launch (incl. its split-K reduce)
            ms = sum(r["ms"] for r in conv)
            flops = sum(r["flops"] for r in conv)
            if gflop_pair:   # the launched layers must add up to the published algorithmic figure (no layer counted twice)
                assert abs(flops / n / 1e9 - gflop_pair) < 1e-3 * gflop_pair, "profile sums to %.4f GFLOP per pair, expected %.3f" % (flops / n / 1e9, gflop_pair)
            else:            # no published per-pair figure for this workload: 2*MAC of the launched conv / deconv / dense layers
                result["gflop_per_pair"] = flops / n / 1e9
                result["pipeline_mfma_frac"] = value / world * flops / n / (PEAK_FP32_MFMA_TFLOPS * 1e12)
            total_ms = sum(r["ms"] for r in recs)
            # per template instance, rocprofv3's convention: the kernel's OWN time (ms - reduce_ms); the conv_splitk_reduce
            # launches that follow some of its launches are listed beside it, not folded in
            by_kernel = {}
            for r in recs:
                e = by_kernel.setdefault(r["kernel"].split("+")[0], {"ms": 0.0, "reduce_ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0, "splitk_launches": 0})
                red = r.get("reduce_ms", 0.0) if "+splitk" in r["kernel"] else 0.0
                e["ms"] += r["ms"] - red; e["reduce_ms"] += red
                e["flops"] += r["flops"]; e["bytes"] += r["bytes"]; e["launches"] += 1
                e["splitk_launches"] += 1 if "+splitk" in r["kernel"] else 0
            rocprof = rocprof_stats()

            def executed_share(tag):
                """MFMA multiply-adds a kernel executes per algorithmic (direct-convolution) multiply-add: 1 for the direct kernels;
                the minimal-filtering kernels of conv_wino.hip compute the same sums with fewer products"""
                import re
                if tag.startswith("wino_deconv"):
                    return 9.0 / 16.0            # F(2,2) x F(2,2) per sub-pixel class
                if tag.startswith("wino3rows<f4"):
                    return 6.0 / 12.0            # F(4,3) row filters
                if tag.startswith("wino3rows<s2"):
                    return 9.0 / 12.0            # stride 2: polyphase F(4,2) + F(4,1) row filters
                if tag.startswith("wino4<t3"):
                    return 6.0 / 12.0            # F(4,3)
                if tag.startswith("wino4<t5"):
                    return 11.0 / 20.0           # polyphase F(4,3) + F(4,2)
                m = re.match(r"(?:wino1d|wino3rows|conv_row<32x128,|wino1d_chain)<?t(\d+)", tag)
                if m:
                    taps = int(m.group(1))
                    return 4.0 / 6.0 if taps == 3 else (taps + 2.0) / (2.0 * taps)   # F(2,3); polyphase F(2,re) + F(2,ro), stride 2
                return 1.0

            def roofline_entry(tag):
                k = by_kernel[tag]
                share = executed_share(tag)
                algorithmic = k["flops"] / (k["ms"] * 1e-3) / 1e12
                achieved = algorithmic * share     # what the matrix pipe executes: <= peak by construction
                e = {
                    "kernel": rocprof_kernel_name(tag), "tag": tag,
                    "bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                    "algorithmic_achieved": algorithmic, "algorithmic_frac": algorithmic / PEAK_FP32_MFMA_TFLOPS,
                    "mfma_flops_executed_per_algorithmic": share,
                    "algorithmic_bytes_per_launch": k["bytes"] / k["launches"], "flops_per_launch": k["flops"] / k["launches"],
                    "launches": k["launches"], "avg_launch_ms": k["ms"] / k["launches"],
                    "kernel_time_share": k["ms"] / total_ms,
                    "splitk_reduce": {"launches": k["splitk_launches"],
                                      "avg_ms": k["reduce_ms"] / k["splitk_launches"] if k["splitk_launches"] else 0.0,
                                      "frac_with_reduce": share * k["flops"] / ((k["ms"] + k["reduce_ms"]) * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS},
                    "timing": "hip events around each launch on the context stream, launches run one after the other (eager), mean of 3 "
                              "passes; kernel only -- the conv_splitk_reduce launch that follows some launches is timed separately (splitk_reduce)",
                }
                e["flops_executed_per_launch"] = share * k["flops"] / k["launches"]
                f = in_flight.get(tag)
                e["regime"] = "stand-alone (one launch at a time: the regime of `value`, one lane)"
                if f and f["ms"] > 0:
                    # the same kernel with `lanes` passes in flight (see in_flight above) -- the regime `value` was measured in, so THESE are
                    # the entry's `achieved` / `frac` / `avg_launch_ms`; the one-launch-at-a-time figures, which are what a
                    # rocprofv3 / PMC pass can see (counters serialise kernels), move to `stand_alone`
                    alg = f["flops"] / (f["ms"] * 1e-3) / 1e12
                    alone = {k: e[k] for k in ("achieved", "frac", "algorithmic_achieved", "algorithmic_frac", "avg_launch_ms", "timing")}
                    e.update(achieved=alg * share, frac=alg * share / PEAK_FP32_MFMA_TFLOPS, algorithmic_achieved=alg, algorithmic_frac=alg / PEAK_FP32_MFMA_TFLOPS,
                             avg_launch_ms=f["ms"] / f["launches"],
                             regime="in flight: %d passes at a time, the regime of `value`" % args.lanes,
                             timing="every launch timed as %d concurrent replays (hipGraphs of 5 launches) on %d streams, hip events around the group; a launch is "
                                    "charged its share 1 / (5 x %d) of the wall time; mean of 2 passes" % (args.lanes, args.lanes, args.lanes))
                    e["stand_alone"] = alone
                    e["in_flight"] = {"lanes": args.lanes, "avg_launch_ms": e["avg_launch_ms"], "achieved": e["achieved"], "frac": e["frac"],
                                      "algorithmic_frac": e["algorithmic_frac"], "note": "(= the entry's own figures; kept under this key for readers of earlier rounds' records)"}
                if share != 1.0:
                    e["note"] = ("minimal-filtering kernel: the matrix pipe executes %.4f of the direct convolution's multiply-adds; `achieved` / `frac` "
                                 "count the executed ones (a roofline fraction, <= 1), `algorithmic_*` price the direct convolution's flops over the same time") % share
                rp = rocprof.get(tag)
                if rp:   # the other clock: rocprofv3 --kernel-trace --stats of this command (graph replay), same kernel sources
                    # (rocprofv3 sees one launch at a time: it is the stand-alone figures this clock must agree with)
                    tgt = e.get("stand_alone", e)
                    tgt["rocprof_avg_launch_ms"] = rp["avg_ms"]
                    tgt["rocprof_calls"] = rp["calls"]
                    tgt["rocprof_frac"] = share * k["flops"] / k["launches"] / (rp["avg_ms"] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS
                    tgt["rocprof_source"] = rp["source"]
                return e

            contraction = {t: k for t, k in by_kernel.items() if k["flops"] > 0}
            dom_tag = max(contraction, key=lambda t: contraction[t]["ms"])
            # `roofline` = the single kernel with the largest share of the pass by its own time (one template instance, as rocprofv3
            # lists it); `roofline_worst` = of the kernels with >= 5 % of the pass, the one furthest below its roofline;
            # `roofline_family` = all contraction launches together
            result["roofline"] = roofline_entry(dom_tag)
            big = [t for t, k in contraction.items() if k["ms"] / total_ms >= 0.05]
            worst_tag = min(big, key=lambda t: executed_share(t) * contraction[t]["flops"] / contraction[t]["ms"]) if big else dom_tag
            result["roofline_worst"] = roofline_entry(worst_tag)
            fam_algorithmic = flops / (ms * 1e-3) / 1e12
            flops_executed = sum(r["flops"] * executed_share(r["kernel"].split("+")[0]) for r in conv)
            fam_achieved = flops_executed / (ms * 1e-3) / 1e12
            result["pipeline_mfma_executed_frac"] = result["pipeline_mfma_frac"] * flops_executed / flops
            result["roofline_family"] = {
                "kernel": "all conv / deconv / dense launches (wino_deconv, wino1d, wino3rows, wino4, conv_frag, conv_frag_chain, conv_stream, conv_stream_chain, conv_patch, deconv4, conv_pair, conv_thin, conv_row, dense_stream, conv_mfma, conv_small kernels)",
                "note": "`achieved` / `frac` count the multiply-adds the matrix pipe executes (the minimal-filtering kernels of conv_wino.hip / conv_row.hip compute the same sums with fewer products); `algorithmic_*` price 2 * MAC of the direct convolutions (BASELINE.md section 2) over the same time",
                "bound": "mfma", "achieved": fam_achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": fam_achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                "algorithmic_achieved": fam_algorithmic, "algorithmic_frac": fam_algorithmic / PEAK_FP32_MFMA_TFLOPS,
                "mfma_flops_executed_per_algorithmic": flops_executed / flops,
                "algorithmic_bytes_per_launch": sum(r["bytes"] for r in conv) / len(conv),
                "launches": len(conv), "splitk_reduce_launches": sum(1 for r in conv if "+splitk" in r["kernel"]),
                "avg_launch_ms": ms / len(conv), "flops_per_launch": flops / len(conv),
                "gflop_per_pair_launched": flops / n / 1e9, "kernel_time_share": ms / total_ms,
            }
            result["roofline_family"]["regime"] = "stand-alone (one launch at a time)"
            if in_flight:
                fms = sum(v["ms"] for t, v in in_flight.items() if v["flops"] > 0)
                fex = sum(v["flops"] * executed_share(t) for t, v in in_flight.items())
                fal = sum(v["flops"] for v in in_flight.values())
                fam = result["roofline_family"]
                fam["stand_alone"] = {k: fam[k] for k in ("achieved", "frac", "algorithmic_achieved", "algorithmic_frac", "avg_launch_ms")}
                fam.update(achieved=fex / (fms * 1e-3) / 1e12, frac=fex / (fms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, algorithmic_achieved=fal / (fms * 1e-3) / 1e12,
                           algorithmic_frac=fal / (fms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, avg_launch_ms=fms / len(conv),
                           regime="in flight: %d passes at a time, the regime of `value`" % args.lanes)
                result["roofline_family"]["in_flight"] = {
                    "lanes": args.lanes, "ms_per_pass": sum(v["ms"] for v in in_flight.values()), "contraction_ms_per_pass": fms,
                    "achieved": fex / (fms * 1e-3) / 1e12, "frac": fex / (fms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                    "algorithmic_frac": fal / (fms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                    "note": "all contraction launches of a pass, each timed as %d concurrent replays (sum of the per-launch shares; compare ms_per_step)" % args.lanes}
            result["kernel_time_shares"] = {k: round(v["ms"] / total_ms, 4) for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1]["ms"])[:8]}
            result["kernel_time_shares"]["conv_splitk_reduce (all)"] = round(sum(v["reduce_ms"] for v in by_kernel.values()) / total_ms, 4)
            # HBM bytes per launch cannot be read from inside the process: they come from the rocprofv3 --pmc passes of
            # this same command (FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 x2 read correction) summarised by
            # tools/pmc_summary.py into profiles/<round>_pmc_summary.json, which records the hash of the kernel sources it
            # measured; a summary taken on other kernels is NOT reported (traffic stays null)
            if args.workload == "full":
                import glob
                from demon_amd import build as hip_build
                pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))
                if pmc:
                    with open(pmc[-1]) as f:
                        summary = json.load(f)
                    if summary.get("csrc_sha") == hip_build.csrc_sha():
                        src = "PMC, %s, kernel sources %s" % (os.path.basename(pmc[-1]), summary["csrc_sha"])
                        c = summary.get("conv", {})
                        if "hbm_traffic_bytes_per_launch" in c:
                            result["roofline_family"]["traffic"] = c["hbm_traffic_bytes_per_launch"]
                            result["roofline_family"]["traffic_unit"] = "bytes per launch (%s)" % src
                            result["roofline_family"]["pmc_mfma_busy_frac"] = c.get("mfma_busy_frac")
                        for key in ("roofline", "roofline_worst"):
                            k = summary.get("kernels", {}).get(result[key]["tag"], {})
                            if "hbm_traffic_bytes_per_launch" in k:
                                result[key]["traffic"] = k["hbm_traffic_bytes_per_launch"]
                                result[key]["traffic_unit"] = "bytes per launch (%s)" % src
                                result[key]["pmc_mfma_busy_frac"] = k.get("mfma_busy_frac")
                    else:
                        result["roofline"]["traffic_note"] = "%s was measured on other kernel sources (%s): not reported" % (
                            os.path.basename(pmc[-1]), summary.get("csrc_sha"))
            if args.layers:
                for r in recs:
                    tf = r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0.0
                    gbs = r["bytes"] / (r["ms"] * 1e-3) / 1e9 if r["ms"] > 0 else 0.0
                    print("%-44s %-30s %8.3f ms %8.2f TF/s %9.1f GB/s" % (r["name"], r["kernel"], r["ms"], tf, gbs), file=sys.stderr)
        if not args.no_e2e and not boot_only:
            # second number of SURVEY 8(d) "GPU timing": host buffers in, host buffers out (H2D of image_pair + image2_2,
            # the whole pipeline, D2H of every output; synchronous pageable copies) -- never `value`
            reps = max(3, min(10, args.steps))
            ctx.full(pair, img2_2, args.iterations)
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.full(pair, img2_2, args.iterations)
            dt = (time.perf_counter() - t0) / reps
            in_b = pair.nbytes + img2_2.nbytes
            out_b = sum(v.nbytes for v in out.values())
            result["extra"] = {"end_to_end_pairs_per_s": n / dt, "end_to_end_ms_per_step": 1e3 * dt,
                               "h2d_bytes_per_step": in_b, "d2h_bytes_per_step": out_b,
                               "note": "demon_full from / to pageable numpy buffers on rank 0, synchronous copies (PCIe-inclusive; not the metric)"}
        if not args.no_cpu_baseline and world == 1 and args.workload == "full":
            os.sched_setaffinity(0, affinity0)        # the baseline gets every host core the process started with
            result["cpu_baseline"] = cpu_baseline(host_weights)
    ctx.close()
    if rank == 0 and not args.no_e2e and args.workload in ("full", "v2"):
        # (after this rank's context is gone: the lanes of the pipeline then get the stream -> hardware-queue mapping of a process of
        # their own, which is what a batch-streaming caller has)
        # the host-to-host rate a batch-streaming caller gets (examples/evaluation.py:225-256 fed in batches): two contexts /
        # streams fed alternately from page-locked host arrays, the copies of one batch under the kernels of the other
        # (demon_amd/pipeline.py); also never `value`
        from demon_amd.pipeline import Pipeline
        pipe = Pipeline(host_weights, batch=n, height=height, width=width, device=local_rank, version=version, contexts=5, calibrate=True)
        try:
            hb = pipe.buffers(8 * n)
            try:
                for i in range(8):
                    hb.image_pair[i * n:(i + 1) * n] = pair
                    hb.image2_2[i * n:(i + 1) * n] = img2_2
                r = pipe.throughput(hb, args.iterations, repeats=2)
                same = all(np.array_equal(hb.out[k][:n], out[k]) for k in hb.out)
            finally:
                hb.release()
        finally:
            pipe.close()
        result["extra"]["pipelined"] = dict(r, outputs_equal_resident_run=bool(same), frac_of_resident=r["pairs_per_s"] / (value / world),
                                            lanes_calibration_pairs_per_s={str(k): round(v, 1) for k, v in (pipe.lane_rates or {}).items()},
                                            note="demon_amd.pipeline.Pipeline: lanes fed round robin from page-locked host arrays, "
                                                 "asynchronous H2D / D2H under the other lanes' kernels; 8 batches per pass")
    if distributed:
        flush_c_stdio()       # RCCL's version banner sits in the C stdio buffer when stdout is a pipe: out with it BEFORE the result,
        dist.barrier()        # on every rank, so that the JSON line is the last thing this job prints
        dist.destroy_process_group()
        flush_c_stdio()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    if "--cpu-pool-leg" in sys.argv:
        _cpu_pool_leg()
    else:
        main()
