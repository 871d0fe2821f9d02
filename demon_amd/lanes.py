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
lanes' resident inputs) and keeps the best prefix of the lanes; it is set-up work like the launch-plan tuning.

Weights exist once per lane (device-to-device copy of the packed slab: demon_copy_weights_from); nothing is shared at run time,
so lanes need no locking: a lane is used by one host thread at a time.
"""
from .engine import DemonContext


class LaneGroup:
    def __init__(self, weights=None, lanes=3, batch=32, height=192, width=256, device=0, version=1, first=None, plan_batch=None):
        """first: an existing context that becomes lane 0 (it already holds its weights, e.g. a rank's context after the RCCL
        broadcast); otherwise lane 0 is created here and takes `weights` (dict tf name -> array)."""
        if lanes < 1:
            raise ValueError("lanes must be >= 1")
        self.batch, self.H, self.W, self.device, self.version = batch, height, width, device, version
        self._owns_first = first is None
        if first is None:
            first = DemonContext(device, batch, height, width, version)
            first.set_weights(weights)
            first.load_tuned_plan(plan_batch or batch, lanes=lanes)
        self.ctxs = [first]
        plan = first.get_plan(plan_batch or batch)
        for _ in range(lanes - 1):
            c = DemonContext(device, batch, height, width, version)
            c.copy_weights_from(first)          # the packed slab, device to device
            if plan:
                c.set_plan(plan_batch or batch, plan)
            self.ctxs.append(c)
        self._side_off = lanes > 1
        if self._side_off:
            for c in self.ctxs:
                c.set_option("side_branches", 0)
        self._next = 0
        self._pads = []
        self.mapping = None

    def __len__(self):
        return len(self.ctxs)

    def close(self):
        for i, c in enumerate(self.ctxs):
            if i or self._owns_first:
                c.close()
            elif self._side_off:
                c.set_option("side_branches", 1)   # a borrowed lane 0 goes back the way it came
        self.ctxs = []
        for p in self._pads:
            p.close()
        self._pads = []

    def _remap(self, pad):
        """every lane gives its HIP streams back, `pad` placeholder streams are created (one-stream contexts that stay alive with the
        group), and the lanes take new streams in order: another stream -> hardware-queue mapping for the same lanes"""
        for c in self.ctxs:
            c.synchronize()
            c.release_streams()
        for p in self._pads:
            p.close()
        self._pads = [DemonContext.ops_only(self.device) for _ in range(pad)]
        for c in self.ctxs:
            c.acquire_streams()

    def _rate(self, k, n, iterations, bootstrap_only, steps_per_lane):
        import time
        best = 0.0
        for _ in range(2):   # (the first round also instantiates graphs / warms caches)
            self.synchronize()
            t0 = time.perf_counter()
            for i in range(steps_per_lane * k):
                c = self.ctxs[i % k]
                c.run_bootstrap(n) if bootstrap_only else c.run_full(n, iterations)
            for c in self.ctxs[:k]:
                c.synchronize()
            best = max(best, n * steps_per_lane * k / (time.perf_counter() - t0))
        return best

    def calibrate(self, n, iterations=3, bootstrap_only=False, steps_per_lane=4, candidates=None, pads=(0, 1, 2, 3)):
        """Measures (inputs must be resident in every lane) the rate of `steps_per_lane * k` steps on the first k lanes for every
        candidate k, under the stream mapping the lanes were created with and -- pads -- after re-creating the lanes' streams behind
        1 .. 3 placeholder streams (the mapping of HIP streams onto hardware queues depends on every stream alive in the process:
        the same lanes measured 3480 .. 4220 pairs/s over 0 .. 3 placeholders, `gpurun_out/r5m/pad.txt`).  Keeps the best
        (placeholders, k), closes the lanes beyond k and returns {"k@placeholders": pairs/s}.  Lane 0 alone (k = 1) runs without
        side branches here, so the comparison is between stream mappings only."""
        ks = [k for k in sorted(set(candidates or range(1, len(self.ctxs) + 1))) if 1 <= k <= len(self.ctxs)]
        rates, best, current = {}, (-1.0, ks[0], 0), 0
        for pad in ([0] + [p for p in pads if p]) if len(self.ctxs) > 1 else [0]:
            if pad != current:
                self._remap(pad)
                current = pad
            for k in ks:
                if pad and k == 1:
                    continue          # one lane does not care where its stream lands
                r = self._rate(k, n, iterations, bootstrap_only, steps_per_lane)
                rates["%d@%d" % (k, pad)] = r
                if r > best[0]:
                    best = (r, k, pad)
        if best[2] != current:
            self._remap(best[2])
        keep = best[1]
        for c in self.ctxs[keep:]:
            c.close()
        del self.ctxs[keep:]
        self._next = 0
        self.mapping = {"lanes": keep, "placeholder_streams": best[2], "pairs_per_s": best[0]}
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

    def run_resident(self, n, steps, iterations=3, bootstrap_only=False):
        """`steps` forward passes over the lanes' resident inputs, round robin; returns without synchronising"""
        k = len(self.ctxs)
        for i in range(steps):
            c = self.ctxs[i % k]
            if bootstrap_only:
                c.run_bootstrap(n)
            else:
                c.run_full(n, iterations)
