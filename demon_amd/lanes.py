"""Lanes: several DemonContexts on ONE GPU, each with its own HIP stream, activation arena and forward hipGraph, fed round robin, so
that several batches are in flight at a time.

Why: a forward pass is ~240 dependent kernel launches of 15-300 us; each of them fills the chip for most of its life but spends
its first and last microseconds (first loads, epilogue, tail of the last wave of workgroups) with idle SIMDs, and the deep layers
run at one or two waves per SIMD.  A second and third pass, independent of the first (other image pairs), fills those gaps: on an
MI355X three lanes of batch 32 run 4176 pairs/s against 3530 on one lane (docs/experiments, "lanes"); two lanes of batch 8 run
2669 against 1865.  The reference's counterpart is a caller that loops over batches (examples/evaluation.py:225-256).

Inside a lane group the "side branches" of a pass (motion head / level-5 head on a second stream of the same context) are switched
off: with several passes in flight they add nothing, and whether the hipGraph runtime maps them onto a hardware queue of their own
or behind another lane's stream depends on the creation order of every stream in the process (measured: 3560-4070 pairs/s for the
same two lanes).  One stream per lane is deterministic.

How many lanes pay off depends on how the HIP runtime maps the lanes' streams onto hardware queues, which in turn depends on
every other stream alive in the process (another context, torch's streams): the same three lanes measured 3660-4176 pairs/s in
two different processes.  `calibrate()` therefore MEASURES the lane counts 1 .. len(group) on the spot (a few steps each, on the
lanes' resident inputs) and keeps the best prefix of the lanes; it is set-up work like the launch-plan tuning.  The measurement
itself lives behind the C ABI (`demon_lanes_calibrate` / `demon_lanes_apply`, include/demon_hip.h; round 5) so that a C / C++
host gets the same behaviour; this class is its Python face.  A winner is remembered per process environment (`mapping_key()`:
device, shape, batch, lane count, WORLD_SIZE, torch loaded or not) in the in-process cache and, when $DEMON_LANES_CACHE names a
JSON file, across processes: `calibrate(reuse=True)` then re-applies it with `demon_lanes_apply` instead of measuring again.

Weights exist once per lane (device-to-device copy of the packed slab: demon_copy_weights_from); nothing is shared at run time,
so lanes need no locking: a lane is used by one host thread at a time.
"""
import os
import sys
import warnings

from .engine import DemonContext

# ---- hardware queues ------------------------------------------------------------------------------------------------------------
# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two busy lanes that share
# one serialise.  Measured on an MI355X at batch 32 (round 5, one box per file):
#   plain process (gpurun_out/r5c/ab.txt):  4 queues -> best 3 lanes, 4 488 - 4 499 pairs/s;  8 queues -> best 4 lanes, 4 622 - 4 640 (+3 %);
#                                           5 / 6 / 10 / 12 / 16 queues and 5 - 6 lanes: no gain
#   under torch.distributed.run with the RCCL communicator alive (gpurun_out/r5h_torchrun_queues.txt): 8 queues -> 3 lanes, 4 452;
#                                           16 queues -> 4 lanes, 4 646
# The runtime reads the variable ONCE, when it initialises (first HIP call of the process).  Until round 5 `import demon_amd` set it for
# everybody; now only this module does, when it is imported -- a process that never runs lanes keeps the runtime's default, and a
# process that imports this module too late (a context or torch.cuda already exists) is TOLD that the request had no effect.
# DEMON_HW_QUEUES=<n> picks another count, DEMON_HW_QUEUES=0 leaves the environment alone; a value the caller exported wins.  A C / C++
# host does the same with demon_hw_queues_hint().
HW_QUEUES = {"requested": None, "env": os.environ.get("GPU_MAX_HW_QUEUES"), "set_by": "caller" if "GPU_MAX_HW_QUEUES" in os.environ else None,
             "runtime_was_up": False}


def _hip_runtime_is_up():
    if DemonContext.created_in_process:
        return True
    torch = sys.modules.get("torch")
    try:
        return bool(torch is not None and torch.cuda.is_initialized())
    except Exception:
        return False


def request_hw_queues():
    """exports GPU_MAX_HW_QUEUES (8, or 16 under a torch.distributed launcher) unless the caller already chose; records what happened
    in HW_QUEUES (every LaneGroup / bench record carries it)"""
    q = os.environ.get("DEMON_HW_QUEUES") or ("16" if ("LOCAL_RANK" in os.environ or "TORCHELASTIC_RUN_ID" in os.environ) else "8")
    HW_QUEUES["requested"] = q
    if q == "0" or "GPU_MAX_HW_QUEUES" in os.environ:
        return HW_QUEUES
    HW_QUEUES["runtime_was_up"] = _hip_runtime_is_up()
    os.environ["GPU_MAX_HW_QUEUES"] = q
    HW_QUEUES["env"], HW_QUEUES["set_by"] = q, "demon_amd.lanes"
    if HW_QUEUES["runtime_was_up"]:
        warnings.warn("demon_amd.lanes: GPU_MAX_HW_QUEUES=%s was exported AFTER the HIP runtime initialised -- it has no effect in this "
                      "process (import demon_amd.lanes, or export the variable, before the first context / torch.cuda call)" % q, RuntimeWarning)
    return HW_QUEUES


request_hw_queues()


def cu_masks(lanes, layout="block", share=1, slots=32, xcds=8):
    """CU masks (lists of 32-bit words) that split the chip between `lanes` lanes, every lane with the same share of EVERY XCD (a mask
    bit i is CU slot i // 8 of XCD i % 8; a slot k is CU k // 4 of shader engine k % 4).  layout "block": lane j owns the slots
    [j * slots / lanes, (j + 1) * slots / lanes) -- two CUs of every shader engine for four lanes; "stride": the slots k with
    k % lanes == j -- one whole shader engine per XCD for four lanes.  share > 1: a lane also gets the slots of the next share - 1
    lanes (overlapping partitions)."""
    per = slots // lanes
    out = []
    for j in range(lanes):
        own = set()
        for d in range(share):
            jj = (j + d) % lanes
            own |= {k for k in range(slots) if (k // per == jj if layout == "block" else k % lanes == jj)}
        bits = 0
        for k in own:
            for x in range(xcds):
                bits |= 1 << (k * xcds + x)
        out.append([(bits >> (32 * w)) & 0xffffffff for w in range(slots * xcds // 32)])
    return out


class LaneGroup:
    _cache = {}     # mapping_key -> {"lanes", "placeholder_streams", "pairs_per_s"}

    def __init__(self, weights=None, lanes=3, batch=32, height=192, width=256, device=0, version=1, first=None, plan_batch=None, partitions=1):
        """first: an existing context that becomes lane 0 (it already holds its weights, e.g. a rank's context after the RCCL
        broadcast); otherwise lane 0 is created here and takes `weights` (dict tf name -> array).
        partitions = P > 1 (round 6): the compute units are split into P equal shares of every XCD (cu_masks) and lane j runs on share
        j % P (demon_set_cu_mask) -- lanes on different shares never compete for a CU; the stream -> hardware-queue mapping is still
        calibrated (two lanes on one queue serialise whatever their masks), but only over placeholder counts: every lane is kept."""
        if lanes < 1:
            raise ValueError("lanes must be >= 1")
        if partitions < 1 or (partitions > 1 and (32 % partitions or lanes % partitions)):
            raise ValueError("partitions must divide 32 CU slots per XCD and the lane count")
        self.batch, self.H, self.W, self.device, self.version = batch, height, width, device, version
        self._owns_first = first is None
        self._plan_batch = plan_batch or batch
        if first is None:
            first = DemonContext(device, batch, height, width, version)
            first.set_weights(weights)
            first.load_tuned_plan(self._plan_batch, lanes=lanes)
        self._first_side = first.get_option("side_branches")     # a borrowed lane 0 goes back the way it came
        self.ctxs = [first]
        plan = first.get_plan(self._plan_batch)
        for _ in range(lanes - 1):
            c = DemonContext(device, batch, height, width, version)
            c.copy_weights_from(first)          # the packed slab, device to device
            if plan:
                c.set_plan(self._plan_batch, plan)
            self.ctxs.append(c)
        self._requested = lanes
        self.partitions = partitions
        if partitions > 1:
            if self._owns_first:
                first.load_tuned_plan(self._plan_batch, lanes=lanes, partitions=partitions)
                plan = first.get_plan(self._plan_batch)
                for c in self.ctxs[1:]:
                    c.clear_plan(self._plan_batch)
                    c.set_plan(self._plan_batch, plan)
            pm = cu_masks(partitions, "block")
            self.set_cu_masks([pm[j % partitions] for j in range(lanes)])
        self._side_off = lanes > 1
        if self._side_off:
            for c in self.ctxs:
                c.set_option("side_branches", 0)
        self._next = 0
        self.mapping = None

    def __len__(self):
        return len(self.ctxs)

    def close(self):
        for i, c in enumerate(self.ctxs):
            if i or self._owns_first:
                c.close()
            else:
                if self.partitions > 1:
                    c.set_cu_mask(None)            # the borrowed context gets the whole chip back
                self._apply(0, ctxs=[c])           # its placeholder streams go; the borrowed context keeps working
                c.set_option("side_branches", self._first_side)
        self.ctxs = []

    # ---- the C ABI underneath ------------------------------------------------------------------------------------------------------
    def _handles(self, ctxs=None):
        import ctypes
        ctxs = self.ctxs if ctxs is None else ctxs
        return (ctypes.c_void_p * len(ctxs))(*[c.h for c in ctxs]), len(ctxs)

    def _apply(self, placeholder_streams, ctxs=None):
        """demon_lanes_apply: every lane gives its HIP streams back, `placeholder_streams` idle streams are created (owned by lane 0)
        and the lanes take new streams in order: another stream -> hardware-queue mapping for the same lanes"""
        arr, k = self._handles(ctxs)
        first = (self.ctxs if ctxs is None else ctxs)[0]
        first._check(first.lib.demon_lanes_apply(arr, k, int(placeholder_streams)))

    def mapping_key(self):
        """what a measured (lanes, placeholder streams) winner depends on, as far as this process can tell: the streams other
        libraries hold (torch.distributed / RCCL under a launcher, torch itself) shift the mapping"""
        return "dev%d_%dx%d_v%d_n%d_l%d_p%d_ws%s_torch%d" % (self.device, self.H, self.W, self.version, self.batch, self._requested, self.partitions,
                                                             os.environ.get("WORLD_SIZE", "1"), int("torch" in sys.modules))

    @classmethod
    def _cache_file(cls):
        import os
        return os.environ.get("DEMON_LANES_CACHE") or None

    @classmethod
    def _cache_load(cls):
        import json
        import os
        path = cls._cache_file()
        if path and os.path.isfile(path):
            try:
                with open(path) as f:
                    for k, v in json.load(f).items():
                        cls._cache.setdefault(k, v)
            except (OSError, ValueError):
                pass

    @classmethod
    def _cache_store(cls):
        """merges this process's winners into the file (other ranks write other keys: re-read just before writing) and replaces it
        atomically, so a concurrent reader never sees a truncated file"""
        import json
        import tempfile
        path = cls._cache_file()
        if path:
            try:
                merged = {}
                if os.path.isfile(path):
                    try:
                        with open(path) as f:
                            merged = json.load(f)
                    except (OSError, ValueError):
                        merged = {}
                merged.update(cls._cache)
                fd, tmp = tempfile.mkstemp(prefix=os.path.basename(path) + ".", suffix=".tmp", dir=os.path.dirname(os.path.abspath(path)))
                with os.fdopen(fd, "w") as f:
                    json.dump(merged, f, indent=1, sort_keys=True)
                os.replace(tmp, path)
            except OSError:
                pass

    def _measure(self, k, n, iterations, bootstrap_only, steps_per_lane):
        """pairs/s of steps_per_lane * k steps round robin over the first k lanes (resident inputs), best of two rounds"""
        import time
        best = 0.0
        for _ in range(2):
            self.synchronize()
            t0 = time.perf_counter()
            for i in range(steps_per_lane * k):
                c = self.ctxs[i % k]
                c.run_bootstrap(n) if bootstrap_only else c.run_full(n, iterations)
            for c in self.ctxs[:k]:
                c.synchronize()
            best = max(best, n * steps_per_lane * k / (time.perf_counter() - t0))
        return best

    def _keep(self, keep, placeholder_streams, rate):
        for c in self.ctxs[keep:]:
            c.close()
        del self.ctxs[keep:]
        self._next = 0
        if keep == 1 and self._side_off:
            # one lane left: it is a plain context again -- side branches back on and the latency plan of its batch size instead of
            # the throughput-mode one (a lone lane on the group's settings is slower than a context that never was in a group)
            self.ctxs[0].set_option("side_branches", self._first_side if not self._owns_first else 1)
            if self._owns_first:
                self.ctxs[0].load_tuned_plan(self._plan_batch, lanes=1)
            self._side_off = False
        self.mapping = {"lanes": keep, "placeholder_streams": placeholder_streams, "pairs_per_s": rate, "hw_queues": dict(HW_QUEUES), "cu_partitions": self.partitions}

    def calibrate(self, n, iterations=3, bootstrap_only=False, steps_per_lane=4, candidates=None, pads=(0, 1, 2, 3, 4, 5), reuse=False):
        """demon_lanes_calibrate (inputs must be resident in every lane): the rate of `steps_per_lane * k` steps on the first k lanes
        for k = 1 .. len(group), under the stream mapping behind 0 .. max(pads) placeholder streams (the mapping of HIP streams onto
        hardware queues depends on every stream alive in the process: the same lanes measured 3480 .. 4220 pairs/s over 0 .. 3
        placeholders, `gpurun_out/r5m/pad.txt`).  Keeps the best (placeholders, k) among the `candidates` lane counts (default:
        all), closes the lanes beyond k and returns {"k@placeholders": pairs/s}.  A group that keeps ONE lane turns it back into a
        plain context (side branches on, latency plan).  reuse: a winner remembered for this process environment (mapping_key())
        is applied and measured once; when its rate is back ({} is returned) the sweep is skipped, else the full calibration runs."""
        from ._lib import LanesResult
        import ctypes
        if self.partitions > 1 and candidates is None:
            candidates = [len(self.ctxs)]      # every partition keeps its lanes: only the stream mapping is searched
        ks = [k for k in sorted(set(candidates or range(1, len(self.ctxs) + 1))) if 1 <= k <= len(self.ctxs)]
        if not ks:
            raise ValueError("candidates %r: no lane count in [1, %d]" % (candidates, len(self.ctxs)))
        key = self.mapping_key()
        if reuse:
            self._cache_load()
            hit = self._cache.get(key)
            if hit and hit["lanes"] in ks:
                # apply the remembered winner and MEASURE it: the placeholder count alone does not reproduce a mapping (see above);
                # only a rate within 2.5 % of the remembered one is taken, anything else falls through to the full calibration
                self._apply(hit["placeholder_streams"])
                got = self._measure(hit["lanes"], n, iterations, bootstrap_only, steps_per_lane)
                if got >= 0.975 * hit["pairs_per_s"]:
                    self._keep(hit["lanes"], hit["placeholder_streams"], hit["pairs_per_s"])
                    self.mapping.update(reused=True, verified_pairs_per_s=got, attempts=1)
                    return {}
        res = LanesResult()
        arr, k = self._handles()
        first = self.ctxs[0]
        mask = 0
        for kk in ks:
            mask |= 1 << kk
        first._check(first.lib.demon_lanes_calibrate(arr, k, int(n), int(iterations), int(bool(bootstrap_only)), int(steps_per_lane),
                                                     int(max(pads) if len(self.ctxs) > 1 and pads else 0), mask, ctypes.byref(res)))
        rates = {}
        for i in range(res.ntable):
            e = res.table[i]
            rates["%d@%d" % (e.lanes, e.placeholder_streams)] = float(e.pairs_per_s)
        # (the C side left the lanes on its winner, applied again and measured again there: `verified`)
        self._keep(res.lanes, res.placeholder_streams, float(res.pairs_per_s))
        self.mapping["verified_pairs_per_s"] = float(res.verified_pairs_per_s)
        self.mapping["attempts"] = int(res.attempts)
        # a winner whose rate never came back when it was applied again (attempts = DEMON_LANES_MAX_ATTEMPTS + 1) is not worth remembering
        self.mapping["reproduced"] = bool(res.verified_pairs_per_s >= 0.975 * res.pairs_per_s)
        if self.mapping["reproduced"] or res.lanes == 1:
            self._cache[key] = {k2: self.mapping[k2] for k2 in ("lanes", "placeholder_streams", "pairs_per_s")}
            self._cache_store()
        return rates

    def next_lane(self):
        c = self.ctxs[self._next]
        self._next = (self._next + 1) % len(self.ctxs)
        return c

    def synchronize(self):
        for c in self.ctxs:
            c.synchronize()

    def upload_inputs(self, batches):
        """batches: one (image_pair, image2_2) per lane; they stay resident in the lanes' input buffers"""
        return [c.upload_inputs(*b) for c, b in zip(self.ctxs, batches)]

    def set_cu_masks(self, masks):
        """demon_set_cu_mask on every lane (masks: one list of 32-bit words per lane, cu_masks(); None / []: every CU again).  Lanes
        on disjoint masks do not compete for compute units wherever the runtime puts their streams."""
        import ctypes
        for c, m in zip(self.ctxs, masks or [[]] * len(self.ctxs)):
            arr = (ctypes.c_uint32 * max(1, len(m)))(*m)
            c._check(c.lib.demon_set_cu_mask(c.h, arr, len(m)))

    def run_group(self, n, launches, iterations=3, bootstrap_only=False):
        """demon_lanes_run_group: `launches` launches of ONE graph that holds a pass of every lane as parallel branches (len(self)
        steps per launch); returns without synchronising (synchronize() on lane 0 waits for all of them)"""
        arr, k = self._handles()
        first = self.ctxs[0]
        for _ in range(launches):
            first._check(first.lib.demon_lanes_run_group(arr, k, int(n), int(iterations), int(bool(bootstrap_only))))

    def run_resident(self, n, steps, iterations=3, bootstrap_only=False):
        """`steps` forward passes over the lanes' resident inputs, round robin; returns without synchronising"""
        k = len(self.ctxs)
        for i in range(steps):
            c = self.ctxs[i % k]
            if bootstrap_only:
                c.run_bootstrap(n)
            else:
                c.run_full(n, iterations)
